#!/usr/bin/env python3
"""Golden vectors for the Moshi-flavour transformer, produced by RUNNING THE REFERENCE's importable twin
tools/tokenizer/MimiCodec/model/modules/transformer.py (byte-identical to the llm_modules/transformer.py
north_star names): whole-sequence forward of
  (mimi)  layer_norm + GELU FFN + interleaved rope + layer_scale + context window  (MimiCodec.py:54-58 flavour)
  (dep)   rms_norm_f32 + silu gating + no positional embedding + weights_per_step  (mllm_model.py:114-143 flavour)
Container-only.  Usage: python tests/golden/make_golden_moshi.py"""
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

from weights import seeded_tensor

MIMI = dict(d_model=64, num_heads=2, num_layers=2, dim_feedforward=128, causal=True, context=20, positional_embedding="rope",
            max_period=10000, norm="layer_norm", layer_scale=0.01, gating="none")
SIN = dict(d_model=64, num_heads=2, num_layers=1, dim_feedforward=128, causal=True, context=None, positional_embedding="sin_rope",
           max_period=10000, positional_scale=0.5, norm="layer_norm", gating="none")
DEP = dict(d_model=64, num_heads=2, num_layers=2, dim_feedforward=96, causal=True, context=None, positional_embedding="none",
           norm="rms_norm_f32", gating="silu", weights_per_step=4)


def moshi_state_dict(shapes, seed):
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        t = seeded_tensor(shp, seed * 6007 + i, std=1.0)
        if k.endswith("scale"):
            t = 0.5 + 0.1 * t              # LayerScale large enough to matter
        elif "norm" in k and (k.endswith("weight") or k.endswith("alpha")):
            t = 1.0 + 0.1 * t
        elif "norm" in k and k.endswith("bias"):
            t = 0.1 * t
        else:
            t = t / shp[-1] ** 0.5
        out[k] = t
    return out


def main():
    from tools.tokenizer.MimiCodec.model.modules.transformer import StreamingTransformer
    out, meta = {}, {}
    for name, cfg, T, seed in (("mimi", MIMI, 50, 51), ("dep", DEP, 4, 52), ("sin", SIN, 20, 53)):
        m = StreamingTransformer(**cfg).eval()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(moshi_state_dict(shapes, seed))
        meta[name + "_keys"] = [[k, list(s)] for k, s in shapes.items()]
        x = seeded_tensor((2, T, cfg["d_model"]), seed + 100, std=1.0)
        with torch.no_grad():
            y = m(x)
        out[name + "_out"] = y.numpy()
        print(name, y.shape, float(y.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "moshi_toy.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "moshi_toy.json"), "w"))


if __name__ == "__main__":
    main()
