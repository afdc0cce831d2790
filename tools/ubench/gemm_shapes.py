#!/usr/bin/env python3
"""Times the large-M linear kernel (prep + gemm) at the prefill / batched-decode shapes of SURVEY.md §8d
configs 3-4 and prints TFLOP/s against the dense bf16 MFMA peak (2.5 PFLOP/s); the row-tiled decode kernel on
the same problem is timed beside it.  Usage on the GPU box: python tools/ubench/gemm_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, PRO_NORM, lib

dev = torch.device("cuda")
dt = torch.bfloat16
L = 4


def packed(N, K):
    return [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]


MS = [int(v) for v in sys.argv[1:]] or [64, 512, 6272]
TRUNK = (("swiglu 3072->2x8192", 8192, 3072, PRO_NORM, EPI_SWIGLU),
         ("down 8192->3072", 3072, 8192, PRO_CAST, EPI_RESIDUAL),
         ("qkv-sized 3072->5120", 5120, 3072, PRO_NORM, EPI_STORE),
         ("oproj 3072->3072", 3072, 3072, PRO_CAST, EPI_RESIDUAL))
DIT = (("dit ff1 1536->6144", 6144, 1536, PRO_CAST, EPI_STORE),      # UA2_SHAPES=dit: the codec DiT's four GEMMs (M = 1000)
       ("dit ff2 6144->1536", 1536, 6144, PRO_CAST, EPI_RESIDUAL),
       ("dit qkv 1536->4608", 4608, 1536, PRO_CAST, EPI_STORE),
       ("dit o 1536->1536", 1536, 1536, PRO_CAST, EPI_RESIDUAL))
DITNORM = (("dit ff1 LN 1536->6144", 6144, 1536, PRO_NORM, EPI_STORE), ("dit qkv LN 1536->4608", 4608, 1536, PRO_NORM, EPI_STORE))
SETS = {"dit": DIT, "ditnorm": DITNORM}
for M in MS:
    ws = ops.linear_workspace(dt, M, 8192, dev)
    for name, N, K, pro, epi in SETS.get(os.environ.get("UA2_SHAPES"), TRUNK):
        w0 = packed(N, K)
        w1 = packed(N, K) if epi == EPI_SWIGLU else [None] * L
        x = torch.randn(M, K, device=dev); nw = torch.ones(K, device=dev)
        y = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
        flop = 2.0 * M * N * K * (2 if epi == EPI_SWIGLU else 1)
        out = []
        for mode in (4, 5, 2):
            if (mode == 2 and M > 512) or (mode == 4 and M > 1024):
                out.append(float("nan")); continue
            lib.ua2_debug_force_general_linear(mode)
            if os.environ.get("UA2_PREPACKED"): pro = PRO_CAST
            pre = os.environ.get("UA2_PREPACKED") and mode == 5   # operand already in fragment order: the GEMM launch alone
            args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, prologue=pro, epilogue=epi, x=None if pre else x, norm_w=nw, y=y,
                               resid=res if epi == EPI_RESIDUAL else None, workspace=None if pre else ws, x_packed=ws if pre else None,
                               launch=False) for a, b in zip(w0, w1)]
            ops.linear_chain_timed(args, 2)
            out.append(ops.linear_chain_timed(args, 5))
        lib.ua2_debug_force_general_linear(0)
        print(f"M={M:5d} {name:22s} skinny {out[0]*1e3:8.1f} us | tiled {out[1]*1e3:8.1f} us {flop/out[1]/1e9:7.1f} TFLOP/s "
              f"({flop/out[1]/1e9/2500*100:4.1f}% of bf16 peak) | row-tiled decode kernel {out[2]*1e3:8.1f} us", flush=True)
        del w0, w1
