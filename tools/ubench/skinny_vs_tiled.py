#!/usr/bin/env python3
"""Batched-decode row counts: the weights-stationary kernel (ua2_skinny.hip, launcher's own choice) against the tiled GEMM on the
model's shapes, pre-packed operand, bits compared.  Where is the cut-over?  Usage on the GPU box: python tools/ubench/skinny_vs_tiled.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, lib

dev = torch.device("cuda")
dt = torch.bfloat16
L = 4
SHAPES = (("trunk qkv-sized", 5120, 3072, EPI_STORE), ("trunk oproj", 3072, 3072, EPI_RESIDUAL),
          ("trunk swiglu", 8192, 3072, EPI_SWIGLU), ("trunk down", 3072, 8192, EPI_RESIDUAL),
          ("dec qkv-sized", 3072, 2048, EPI_STORE), ("dec oproj", 2048, 2048, EPI_RESIDUAL),
          ("dec swiglu", 8192, 2048, EPI_SWIGLU), ("dec down", 2048, 8192, EPI_RESIDUAL),
          ("projection", 2048, 3072, EPI_STORE), ("audio_head", 12296, 2048, EPI_STORE))
for M in [int(v) for v in sys.argv[1:]] or [64, 128, 192, 256, 320]:
    tot = {"skinny2": 0.0, "tiled": 0.0}
    for name, N, K, epi in SHAPES:
        w0 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]
        w1 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)] if epi == EPI_SWIGLU else [None] * L
        x = torch.randn(M, K, device=dev); res = torch.randn(M, N, device=dev)
        ws = ops.linear_workspace(dt, M, K, dev)
        y0 = torch.empty(M, N, device=dev)
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w0[0], w1=w1[0], prologue=PRO_CAST, epilogue=epi, x=x, y=y0,
                   resid=res if epi == EPI_RESIDUAL else None, workspace=ws)
        torch.cuda.synchronize()
        row = {}
        for v in ("skinny2", "tiled"):
            if v == "tiled":
                os.environ["UA2_SKINNY_MAX_ROWS"] = "0"; os.environ["UA2_SKINNY2"] = "off"
            else:
                os.environ["UA2_SKINNY_MAX_ROWS"] = "100000"; os.environ.pop("UA2_SKINNY2", None)
            y = torch.zeros(M, N, device=dev)
            args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, prologue=PRO_CAST, epilogue=epi, x_packed=ws, y=y,
                               resid=res if epi == EPI_RESIDUAL else None, launch=False) for a, b in zip(w0, w1)]
            ops.linear_chain_timed(args[:1], 1)
            torch.cuda.synchronize()
            same = bool(torch.equal(y, y0))
            ops.linear_chain_timed(args, 3)
            row[v] = (ops.linear_chain_timed(args, 10) * 1e3, same)
            tot[v] += row[v][0]
        print(f"M={M:4d} {name:16s} skinny2 {row['skinny2'][0]:7.1f} us{'' if row['skinny2'][1] else ' DIFF'} | tiled {row['tiled'][0]:7.1f} us{'' if row['tiled'][1] else ' DIFF'}"
              f" | {'TILED' if row['tiled'][0] < row['skinny2'][0] else 'skinny2'}", flush=True)
        del w0, w1
    print(f"M={M:4d} sum skinny2 {tot['skinny2']:.1f} tiled {tot['tiled']:.1f}", flush=True)
