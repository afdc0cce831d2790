#!/usr/bin/env python3
"""A/B of the order-free GEMM's cost choices (same bits): kernel variant (UA2_GEMM2_VAR), tile (UA2_GEMM2_BMT), ring (UA2_GEMM2_NB)
per shape, interleaved in ONE process (the launcher reads the hooks per launch).  The variants beyond the production pair need a library
built with -DUA2_G2_EXPERIMENTS (tools/ubench/build_alt.sh g2x ua2_gemm2.hip -DUA2_G2_EXPERIMENTS; UA2_LIB=...).  GEMM launch alone, four rotating weight sets.
Usage: python tools/ubench/gemm2_variants.py            (UA2_LIB=<knock-out build> for the timing-only builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_GELU, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, SUM_ORDER_FREE

dev, dt, L = torch.device("cuda"), torch.bfloat16, 4
SHAPES = [("swiglu", 6272, 8192, 3072, EPI_SWIGLU), ("qkv", 6272, 5120, 3072, EPI_STORE), ("down", 6272, 3072, 8192, EPI_RESIDUAL),
          ("oproj", 6272, 3072, 3072, EPI_RESIDUAL), ("swiglu1k", 1024, 8192, 3072, EPI_SWIGLU), ("down1k", 1024, 3072, 8192, EPI_RESIDUAL),
          ("dit-qkv", 1000, 4608, 1536, EPI_STORE), ("dit-o", 1000, 1536, 1536, EPI_RESIDUAL), ("dit-ff1", 1000, 6144, 1536, EPI_GELU),
          ("dit-ff2", 1000, 1536, 6144, EPI_RESIDUAL), ("dit8-qkv", 8000, 4608, 1536, EPI_STORE), ("dit8-o", 8000, 1536, 1536, EPI_RESIDUAL),
          ("dit8-ff2", 8000, 1536, 6144, EPI_RESIDUAL)]
VARIANTS = [("inv", None), ("auto", dict()), ("b16", dict(UA2_GEMM2_BMT=16)), ("b8", dict(UA2_GEMM2_BMT=8)),
            ("b16v0", dict(UA2_GEMM2_BMT=16, UA2_GEMM2_VAR=0)), ("b16v1", dict(UA2_GEMM2_BMT=16, UA2_GEMM2_VAR=1)), ("b8n6", dict(UA2_GEMM2_BMT=8, UA2_GEMM2_NB=6))]
if os.environ.get("UA2_VARIANTS"):
    keep = os.environ["UA2_VARIANTS"].split(",")
    VARIANTS = [v for v in VARIANTS if v[0] in keep]
HOOKS = ("UA2_GEMM2_BMT", "UA2_GEMM2_VAR", "UA2_GEMM2_NB")
only = os.environ.get("UA2_ONLY", "").split(",") if os.environ.get("UA2_ONLY") else None
print("shape".ljust(10), " ".join(f"{n:>9s}" for n, _ in VARIANTS), " (us; TF of the best)")
for name, M, N, K, epi in SHAPES:
    if only and name not in only:
        continue
    w0 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]
    w1 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)] if epi == EPI_SWIGLU else [None] * L
    xp = torch.randn((M + 15) // 16 * 16 * K, device=dev).to(dt)
    y = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
    flop = 2.0 * M * N * K * (2 if epi == EPI_SWIGLU else 1)
    times = {n: [] for n, _ in VARIANTS}
    for rnd in range(3):
        for n, env in VARIANTS:
            for h in HOOKS:
                os.environ.pop(h, None)
            for k, v in (env or {}).items():
                os.environ[k] = str(v)
            args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, w1=b, prologue=PRO_CAST, epilogue=epi, x_packed=xp, y=y, resid=res if epi == EPI_RESIDUAL else None,
                               sum_order=0 if env is None else SUM_ORDER_FREE, launch=False) for a, b in zip(w0, w1)]
            if rnd == 0:
                ops.linear_chain_timed(args, 1)
            times[n].append(ops.linear_chain_timed(args, 5))
    best = min(min(v) for v in times.values())
    print(name.ljust(10), " ".join(f"{min(times[n]) * 1e3:9.1f}" for n, _ in VARIANTS), f"  {flop / best / 1e9:7.1f} TF", flush=True)
    del w0, w1, xp, y, res
