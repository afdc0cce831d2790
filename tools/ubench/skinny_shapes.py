#!/usr/bin/env python3
"""Sweep of the batched-decode linear kernel (csrc/ua2_skinny.hip) over its tile parameters at the model's shapes,
beside round 2's skinny kernel and the tiled GEMM; every variant's output is compared bit for bit with the old kernel's.
The operand is handed over pre-packed (x_packed), so the numbers are the GEMM launch alone (no prep launch).
Usage on the GPU box: python tools/ubench/skinny_shapes.py [M ...]"""
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
          ("projection", 2048, 3072, EPI_STORE), ("audio_head", 12296, 2048, EPI_STORE), ("lm_head", 128256, 3072, EPI_STORE))
MS = [int(v) for v in sys.argv[1:]] or [64, 256]


def variants(M):
    mt = (M + 15) // 16
    out = ["off", "auto"]
    for ct in (1, 2):
        for mtw, la in ((4, 1), (4, 2), (2, 2)):
            ps = sorted({1, max(1, (mt + mtw - 1) // mtw), max(1, (mt + 2 * mtw - 1) // (2 * mtw)), max(1, (mt + 4 * mtw - 1) // (4 * mtw))})
            for p in ps:
                out.append(f"{ct},{mtw},{la},{p}")
    # round 6: weight-ring forms (one pass, grid.y = row tiles / mt): "ct,mt,la,1,wd"
    out += ["2,4,1,1,6", "2,4,2,1,6", "2,4,1,1,4", "2,4,1,1,3", "2,2,2,1,4",                       # SwiGLU
            "2,2,2,1,8", "2,4,1,1,8", "3,2,2,1,4", "4,2,2,1,4",                                   # RESIDUAL (+ 2,2,2,1,4 / 2,4,1,1,4)
            "3,2,2,1,6", "4,2,2,1,6", "4,2,2,1,4", "4,4,1,1,4", "3,4,1,1,6", "3,4,1,1,4"]         # STORE / q|k|v
    out += ["1,4,4,1,0,2", "1,4,1,1,0,2", "1,2,4,2,0,2", "1,2,4,1,0,2"]      # two ranges per wave (K = 8192)
    return out


for M in MS:
    for name, N, K, epi in SHAPES:
        if N > 100000 and M > 64:
            Lw = 1
        else:
            Lw = L if N < 100000 else 2
        w0 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(Lw)]
        w1 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(Lw)] if epi == EPI_SWIGLU else [None] * Lw
        x = torch.randn(M, K, device=dev)
        res = torch.randn(M, N, device=dev)
        ws = ops.linear_workspace(dt, M, K, dev)
        # one ordinary launch fills the workspace with the packed operand; reuse it as x_packed
        os.environ["UA2_SKINNY_MAX_ROWS"] = "100000"
        os.environ["UA2_SKINNY2"] = "off"
        y0 = torch.empty(M, N, device=dev)
        lib.ua2_debug_force_general_linear(4)
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w0[0], w1=w1[0], prologue=PRO_CAST, epilogue=epi, x=x, y=y0,
                   resid=res if epi == EPI_RESIDUAL else None, workspace=ws)
        lib.ua2_debug_force_general_linear(0)
        torch.cuda.synchronize()
        results = []
        for v in variants(M) + ["tiled"]:
            if v == "tiled":
                os.environ["UA2_SKINNY_MAX_ROWS"] = "0"
                os.environ["UA2_SKINNY2"] = "off"
            elif v == "auto":
                os.environ["UA2_SKINNY_MAX_ROWS"] = "100000"
                os.environ.pop("UA2_SKINNY2", None)
            else:
                os.environ["UA2_SKINNY_MAX_ROWS"] = "100000"
                os.environ["UA2_SKINNY2"] = v
            y = torch.zeros(M, N, device=dev)
            try:
                args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, prologue=PRO_CAST, epilogue=epi, x_packed=ws, y=y,
                                   resid=res if epi == EPI_RESIDUAL else None, launch=False) for a, b in zip(w0, w1)]
                ops.linear_chain_timed(args[:1], 1)
                torch.cuda.synchronize()
                same = bool(torch.equal(y, y0))
                ops.linear_chain_timed(args, 3)
                t = ops.linear_chain_timed(args, 10) * 1e3
            except Exception as e:      # variant not instantiated for this geometry
                continue
            results.append((t, v, same))
        base = [r for r in results if r[1] == "off"][0][0]
        tl = [r for r in results if r[1] == "tiled"]
        results.sort()
        wbytes = N * K * 2 * (2 if epi == EPI_SWIGLU else 1)
        best = ", ".join(f"{v}:{t:.1f}{'' if same else '(DIFF)'}" for t, v, same in results[:8])
        bad = [v for _, v, same in results if not same]
        print(f"M={M:4d} {name:16s} N={N:6d} K={K:5d} old {base:7.1f} us | tiled {tl[0][0] if tl else float('nan'):7.1f} | best: {best} | "
              f"W/best {wbytes/results[0][0]/1e6:.2f} TB/s | mismatches: {bad}", flush=True)
        del w0, w1
