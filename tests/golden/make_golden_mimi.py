#!/usr/bin/env python3
"""Golden vectors for the MimiCodec assembly, produced by RUNNING THE REFERENCE
(tools/tokenizer/MimiCodec/model/models/MimiCodec.py, importable as-is with NO_TORCH_COMPILE=1):
a toy-sized instance (hop 16, stride-2 learnt down-sampler with replicate padding, channel-wise learnt up-sampler,
2-layer context-10 transformers, 1 + 3 codebooks of 64 x 8) on synthetic weights:
  wav (2, 1, 640) -> latent before the quantizer -> codes (2, 4, 20) -> de-quantised latent -> up-sampled -> wav.
Container-only.  Usage: python tests/golden/make_golden_mimi.py"""
import json
import os
import sys

os.environ.setdefault("NO_TORCH_COMPILE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from weights import mimi_state_dict, seeded_tensor

TOY_MIMI = dict(sample_rate=24000, n_filters=4, encoder_rates=[2, 2, 2, 2], compress=2, causal=True, latent_dim=64,
                codebook_size=64, codebook_dim=8, rvq_layers=4, num_heads=2, num_layers=2, layer_scale=0.01, context=10,
                semantic_feature_dim=16, target_frame_rate=750.0)


def main():
    from tools.tokenizer.MimiCodec.model.models.MimiCodec import MimiCodec
    torch.manual_seed(0)
    m = MimiCodec(**TOY_MIMI).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(mimi_state_dict(shapes, 77))
    wav = seeded_tensor((2, 1, 640), 4321, std=0.3)
    out = {}
    with torch.no_grad():
        z = m.downsample(m.encoder_transformer(m.encoder(wav))[0])
        codes = m.encode(wav)
        zq = m.quantizer.decode(codes)
        up = m.upsample(zq)
        rec = m.decode(codes)
        # margins of the nearest-codeword decisions of the first level of each half (near-ties may flip in fp32)
        out["latent"], out["codes"], out["zq"], out["up"], out["rec"] = z.numpy(), codes.numpy(), zq.numpy(), up.numpy(), rec.numpy()
    np.savez_compressed(os.path.join(HERE, "mimi_toy.npz"), **out)
    json.dump(dict(config=TOY_MIMI, keys=[[k, list(s)] for k, s in shapes.items()]), open(os.path.join(HERE, "mimi_toy.json"), "w"))
    print({k: v.shape for k, v in out.items()}, "rec rms", float(np.sqrt((out["rec"] ** 2).mean())))


if __name__ == "__main__":
    main()
