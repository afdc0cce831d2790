#!/usr/bin/env python3
"""The order-free GEMM (ua2_gemm2.hip) beside the invariant tiled kernel (ua2_gemm.hip) at the shapes that matter: the LM's four
Linear layers at prefill row counts, the DiT's four at one window (1000 rows) and eight (8000).  GEMM launch alone (operand already
in fragment order), four rotating weight sets, HIP-event timed (ua2_linear_chain_timed).
Usage on the GPU box: python tools/ubench/gemm2_shapes.py [M ...]   (env UA2_SHAPES=dit|trunk, UA2_GEMM2_* hooks apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_GELU, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, SUM_ORDER_FREE, lib

dev, dt, L = torch.device("cuda"), torch.bfloat16, 4
TRUNK = (("swiglu 3072->2x8192", 8192, 3072, EPI_SWIGLU), ("down 8192->3072", 3072, 8192, EPI_RESIDUAL),
         ("qkv-sized 3072->5120", 5120, 3072, EPI_STORE), ("oproj 3072->3072", 3072, 3072, EPI_RESIDUAL))
DIT = (("dit qkv 1536->4608", 4608, 1536, EPI_STORE), ("dit o 1536->1536", 1536, 1536, EPI_RESIDUAL),
       ("dit ff1 1536->6144", 6144, 1536, EPI_GELU), ("dit ff2 6144->1536", 1536, 6144, EPI_RESIDUAL))
which = os.environ.get("UA2_SHAPES", "both")
sets = [("trunk", TRUNK, [2048, 6272])] * (which in ("both", "trunk")) + [("dit", DIT, [1000, 8000])] * (which in ("both", "dit"))
for tag, shapes, default_ms in sets:
    for M in ([int(v) for v in sys.argv[1:]] or default_ms):
        for name, N, K, epi in shapes:
            w0 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]
            w1 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)] if epi == EPI_SWIGLU else [None] * L
            xp = (torch.randn((M + 15) // 16 * 16 * K, device=dev)).to(dt)           # any bits will do for timing: random bf16 operand
            y = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
            sw = torch.empty(max(4 * M * N if M <= 2048 else 0, 16 << 20), device=dev) if epi == EPI_RESIDUAL else None   # K slabs on small grids; the tail split of a thin last round (round 6)
            flop = 2.0 * M * N * K * (2 if epi == EPI_SWIGLU else 1)
            out = []
            lib.ua2_debug_force_general_linear(5)
            for order in (0, SUM_ORDER_FREE):
                args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, prologue=PRO_CAST, epilogue=epi, x_packed=xp, y=y,
                                   resid=res if epi == EPI_RESIDUAL else None, split_ws=sw, sum_order=order, launch=False) for a, b in zip(w0, w1)]
                ops.linear_chain_timed(args, 2)
                out.append(min(ops.linear_chain_timed(args, 5) for _ in range(3)))
            lib.ua2_debug_force_general_linear(0)
            print(f"M={M:5d} {name:22s} invariant {out[0]*1e3:8.1f} us {flop/out[0]/1e9:7.1f} TF | order-free {out[1]*1e3:8.1f} us {flop/out[1]/1e9:7.1f} TF "
                  f"({flop/out[1]/1e9/2500*100:4.1f}% of bf16 peak)  x{out[0]/out[1]:.2f}", flush=True)
            del w0, w1, xp, y, res
