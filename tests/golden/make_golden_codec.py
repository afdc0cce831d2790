#!/usr/bin/env python3
"""Golden vectors for the codec conv stacks, produced by RUNNING THE REFERENCE:
  * tools/tokenizer/ReasoningCodec_film/models/scalar24k.py  ScalarModel.encode / .decode  (live codec)
  * tools/tokenizer/MimiCodec/model/modules/seanet.py        SEANetEncoder / SEANetDecoder (named family)
at toy sizes, seeded weights (tests/golden/weights.py; weight_g drawn around 1 so weight-norm matters).
Container-only (needs /root/reference; torchaudio / pytorch_lightning / omegaconf stubbed as empty
modules).  Only tests/golden/codec_toy.npz + .json travel.
Usage: python tests/golden/make_golden_codec.py
"""
import json
import os
import sys
import types

os.environ.setdefault("NO_TORCH_COMPILE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from weights import seeded_tensor

SCALAR_CFG = dict(num_bands=1, sample_rate=24000, causal=True, num_samples=2, downsample_factors=[2, 4],
                  downsample_kernel_sizes=[4, 8], upsample_factors=[4, 2], upsample_kernel_sizes=[8, 4],
                  latent_hidden_dim=24, default_kernel_size=7, delay_kernel_size=5, init_channel=8, res_kernel_size=7)
SEANET_CFG = dict(channels=1, dimension=32, causal=True, n_filters=8, n_residual_layers=1, activation="ELU", compress=2,
                  dilation_base=2, disable_norm_outer_blocks=0, kernel_size=7, residual_kernel_size=3, last_kernel_size=3,
                  norm="none", pad_mode="constant", ratios=[4, 2], true_skip=True)


def codec_state_dict(shapes, seed):
    """Seeded fp32 weights: conv filters ~ N(0, 1/fan_in), weight_g around 1, PReLU slopes around 0.25."""
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        t = seeded_tensor(shp, seed * 7919 + i, std=1.0)
        if k.endswith("weight_g"):
            t = 1.0 + 0.2 * t
        elif "activation" in k:
            t = 0.25 + 0.05 * t
        elif k.endswith("bias"):
            t = 0.1 * t
        else:
            fan = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            t = t / max(fan, 1) ** 0.5
        out[k] = t
    return out


def main():
    for name in ("torchaudio", "pytorch_lightning", "omegaconf"):
        m = types.ModuleType(name)
        sys.modules.setdefault(name, m)
    sys.modules["pytorch_lightning"].LightningModule = torch.nn.Module
    sys.modules["omegaconf"].OmegaConf = object
    from tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    from tools.tokenizer.MimiCodec.model.modules.seanet import SEANetDecoder, SEANetEncoder
    out, meta = {}, {}
    torch.set_num_threads(4)

    m = ScalarModel(**SCALAR_CFG).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(codec_state_dict(shapes, 21))
    meta["scalar_keys"] = [[k, list(s)] for k, s in shapes.items()]
    wav = seeded_tensor((2, 1, 16 * 130 + 5), 31, std=0.3)          # ragged: not a multiple of the hop (16)
    with torch.no_grad():
        lat = m.encode(wav)
        rec = m.decode(lat)
    out.update(scalar_latent=lat.numpy(), scalar_wav=rec.numpy())

    enc, dec = SEANetEncoder(**SEANET_CFG).eval(), SEANetDecoder(**SEANET_CFG).eval()
    for tag, mod, seed in (("seanet_enc", enc, 41), ("seanet_dec", dec, 42)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict(codec_state_dict(shapes, seed))
        meta[tag + "_keys"] = [[k, list(s)] for k, s in shapes.items()]
    wav2 = seeded_tensor((2, 1, 8 * 77 + 3), 32, std=0.3)
    with torch.no_grad():
        z = enc(wav2)
        y = dec(z)
    out.update(seanet_latent=z.numpy(), seanet_wav=y.numpy())
    np.savez_compressed(os.path.join(HERE, "codec_toy.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "codec_toy.json"), "w"))
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
