#!/usr/bin/env python3
"""A/B of the scaled-norm consumer in isolation: the same packed operand through UA2_PRO_CAST and UA2_PRO_SCALED (row scales from
sum-of-squares partials) at the model's consumer shapes, skinny2's own variant choice.  Usage: python tools/ubench/scaled_ab.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_STORE, EPI_SWIGLU, PRO_CAST, lib

PRO_SCALED = 4
dev, dt, L = torch.device("cuda"), torch.bfloat16, 4
for M in [int(v) for v in sys.argv[1:]] or [64, 256]:
    for name, N, K, epi in (("trunk qkv-sized", 5120, 3072, EPI_STORE), ("trunk swiglu", 8192, 3072, EPI_SWIGLU),
                            ("dec qkv-sized", 3072, 2048, EPI_STORE), ("dec swiglu", 8192, 2048, EPI_SWIGLU), ("audio_head", 12296 // 16 * 16, 2048, EPI_STORE)):
        w0 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]
        w1 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)] if epi == EPI_SWIGLU else [None] * L
        x = torch.randn(M, K, device=dev)
        ws = ops.linear_workspace(dt, M, K, dev)
        y = torch.empty(M, N, device=dev)
        os.environ["UA2_SKINNY_MAX_ROWS"] = "100000"
        lib.ua2_debug_force_general_linear(4)
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w0[0], w1=w1[0], prologue=PRO_CAST, epilogue=epi, x=x, y=y, workspace=ws)   # fills ws with the packed operand
        lib.ua2_debug_force_general_linear(0)
        ssq = (torch.rand(M, K // 16, device=dev) * 16).contiguous()
        out = {}
        for tag, kw in (("cast", dict(prologue=PRO_CAST)), ("scaled", dict(prologue=PRO_SCALED, x_ssq=ssq, eps=1e-5))):
            args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, epilogue=epi, x_packed=ws, y=y, launch=False, **kw) for a, b in zip(w0, w1)]
            ops.linear_chain_timed(args, 3)
            out[tag] = ops.linear_chain_timed(args, 20) * 1e3
        line = f"M={M:4d} {name:16s} cast {out['cast']:6.1f} us   scaled {out['scaled']:6.1f} us   (+{out['scaled'] - out['cast']:.1f})"
        if epi == EPI_SWIGLU:                    # round 6: the weight-ring forms (ct,mt,la,passes,wd) of the scaled consumer, bits checked against auto
            args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, epilogue=epi, x_packed=ws, y=y, launch=False, prologue=PRO_SCALED, x_ssq=ssq, eps=1e-5) for a, b in zip(w0, w1)]
            ops.linear_chain_timed(args[:1], 1); torch.cuda.synchronize(); ref = y.clone()
            for v in ("2,4,1,1,6", "2,4,1,1,4", "2,4,1,1,3", "2,4,2,1,6"):
                os.environ["UA2_SKINNY2"] = v
                try:
                    y.zero_(); ops.linear_chain_timed(args[:1], 1); torch.cuda.synchronize()
                    same = bool(torch.equal(y, ref))
                    ops.linear_chain_timed(args, 3)
                    line += f"  | {v}: {ops.linear_chain_timed(args, 20) * 1e3:.1f}{'' if same else ' DIFF'}"
                except Exception:
                    line += f"  | {v}: n/a"
                os.environ.pop("UA2_SKINNY2", None)
        print(line, flush=True)
