#!/usr/bin/env python3
"""Golden vectors for the RVQ search/lookup, produced by RUNNING THE REFERENCE's vendored
tools/tokenizer/MimiCodec/model/quantization/core_vq.py (ResidualVectorQuantization.encode / .decode).
Container-only (needs /root/reference); only tests/golden/rvq_*.npz travels.

Inputs and codebooks are regenerated from seeds by the tests (tests/golden/weights.py); the file holds
the reference's outputs: codes (L, B, T) and decode() output, for
  (a) the Mimi-like flavour: codebook_dim == dim (no projections inside the layers), D=32, C=512, L=4
  (b) the same with L=8, D=64, C=256 and exact ties (duplicated codewords) to pin the first-index rule.
Usage: python tests/golden/make_golden_rvq.py
"""
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

CASES = dict(a=dict(D=32, C=512, L=4, B=3, T=40, seed=11, dup=False),
             b=dict(D=64, C=256, L=8, B=2, T=25, seed=12, dup=True))


def make_inputs(c):
    x = seeded_tensor((c["B"], c["D"], c["T"]), c["seed"], std=1.0)
    emb = seeded_tensor((c["L"], c["C"], c["D"]), c["seed"] + 1000, std=1.0)
    # later levels quantise residuals, which are smaller: shrink their codebooks like a trained RVQ
    for l in range(c["L"]):
        emb[l] *= 0.6 ** l
    if c["dup"]:
        emb[:, 7] = emb[:, 3]           # exact duplicates: argmin must return the lower index
        emb[:, 100] = emb[:, 50]
    return x, emb


def main():
    from tools.tokenizer.MimiCodec.model.quantization.core_vq import ResidualVectorQuantization
    out = {}
    for name, c in CASES.items():
        x, emb = make_inputs(c)
        rvq = ResidualVectorQuantization(num_quantizers=c["L"], codebook_offset=0, dim=c["D"], codebook_size=c["C"])
        for l, layer in enumerate(rvq.layers):
            cb = layer._codebook
            cb.embedding_sum.copy_(emb[l])       # cluster_usage stays 1 -> embedding == embedding_sum (core_vq.py:143-150)
            cb._initialized.fill_(1.0)
        rvq.eval()
        with torch.no_grad():
            codes = rvq.encode(x)                 # (L, B, T)
            dec = rvq.decode(codes)               # (B, D, T)
        out[f"{name}_codes"] = codes.numpy().astype(np.int32)
        out[f"{name}_decoded"] = dec.numpy()
        print(name, codes.shape, dec.shape, "dup hits:", int(((codes == 7) | (codes == 100)).sum()))
    np.savez_compressed(os.path.join(HERE, "rvq_toy.npz"), **out)


if __name__ == "__main__":
    main()
