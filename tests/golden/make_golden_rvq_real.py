#!/usr/bin/env python3
"""Real-size RVQ golden: codes produced by RUNNING THE REFERENCE's vendored
tools/tokenizer/MimiCodec/model/quantization/core_vq.py (`ResidualVectorQuantization.encode`, :365-376, whose
`_quantize` :179-185 measures distances with `torch.cdist` — a GEMM expansion |x|^2 - 2 x.e + |e|^2) at the live codec's
codebook sizes (AudioDiffusion1D.py:183-187,256-264): 6 levels x 8192 x 32 and 8 levels x 4096 x 64, N = 2048 vectors.

The oracle (oracle/rvq_oracle.c) and the HIP kernel measure d2 = sum_k fma(x_k - e_k, x_k - e_k, .) directly.  The two
forms can disagree where two codewords are equidistant within fp32 expansion noise.  Seeds are NOT chosen to avoid that:
the rows where the reference's codes differ from the direct form's are RECORDED here (row, first differing level, both
codes, the direct-form distance gap between the two candidates) and asserted as an exact list by the tests; every other
row must be equal on all levels.

Container-only (needs /root/reference); inputs and codebooks are regenerated from seeds by the tests
(tests/golden/weights.py::seeded_tensor), the file holds the reference's codes and the mismatch records.
Usage: python tests/golden/make_golden_rvq_real.py
"""
import os
import sys

os.environ.setdefault("NO_TORCH_COMPILE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from weights import seeded_tensor

REAL_CASES = dict(acoustic=dict(D=32, C=8192, L=6, N=2048, seed=21),     # rvq_acoustic: 6 x 8192 x 32 (AudioDiffusion1D.py:256-264)
                  semantic=dict(D=64, C=4096, L=8, N=2048, seed=22))     # 8 x 4096 x 64 (:183-187)


def make_real_inputs(c):
    x = seeded_tensor((c["N"], c["D"]), c["seed"], std=1.0)
    emb = seeded_tensor((c["L"], c["C"], c["D"]), c["seed"] + 1000, std=1.0)
    for l in range(c["L"]):
        emb[l] *= 0.7 ** l              # later levels quantise smaller residuals
    return x, emb


def main():
    sys.path.insert(0, "/root/reference")
    from tools.tokenizer.MimiCodec.model.quantization.core_vq import ResidualVectorQuantization
    from oracle import rvq_oracle
    torch.set_num_threads(8)
    out = {}
    for name, c in REAL_CASES.items():
        x, emb = make_real_inputs(c)
        rvq = ResidualVectorQuantization(num_quantizers=c["L"], codebook_offset=0, dim=c["D"], codebook_size=c["C"])
        for l, layer in enumerate(rvq.layers):
            cb = layer._codebook
            cb.embedding_sum.copy_(emb[l])       # cluster_usage stays 1 -> embedding == embedding_sum (core_vq.py:143-150)
            cb._initialized.fill_(1.0)
        rvq.eval()
        with torch.no_grad():
            codes = rvq.encode(x.t()[None].contiguous())          # (L, 1, N): the layer works on (B, D, T)
        ref = np.ascontiguousarray(codes[:, 0].numpy().T).astype(np.int32)      # (N, L)
        o_codes, _, margin = rvq_oracle.rvq_encode(x.numpy(), emb.numpy(), want_margin=True)
        rows = np.nonzero((ref != o_codes).any(1))[0]
        rec = []
        for r in rows:
            l = int(np.nonzero(ref[r] != o_codes[r])[0][0])
            # the residual both forms saw at level l (identical codes before it), and the direct-form gap between the two candidates
            res = x[r].double().clone()
            for k in range(l):
                res -= emb[k, int(ref[r, k])].double()
            da = float(((res - emb[l, int(ref[r, l])].double()) ** 2).sum())
            db = float(((res - emb[l, int(o_codes[r, l])].double()) ** 2).sum())
            rec.append((int(r), l, int(ref[r, l]), int(o_codes[r, l]), abs(da - db) / max(da, db)))
        out[f"{name}_codes"] = ref.astype(np.int16)
        out[f"{name}_mismatch"] = np.array([[a, b, c_, d] for a, b, c_, d, _ in rec], np.int32).reshape(-1, 4)
        out[f"{name}_mismatch_relgap"] = np.array([e for *_, e in rec], np.float64)
        out[f"{name}_min_margin"] = np.array(float(margin.min()), np.float64)     # smallest top-2 distance gap the direct form saw (absolute)
        print(name, ref.shape, "rows where cdist-argmin != direct-form argmin:", len(rec), "of", c["N"],
              "(searches:", c["N"] * c["L"], ") smallest top-2 gap:", float(margin.min()), [(a, b, f"{e:.1e}") for a, b, _, _, e in rec])
    np.savez_compressed(os.path.join(HERE, "rvq_real.npz"), **out)


if __name__ == "__main__":
    main()
