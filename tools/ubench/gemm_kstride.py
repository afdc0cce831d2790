#!/usr/bin/env python3
"""Does the tiled GEMM's time depend on the fragment-stream stride (K / 32 KiB per 16-row tile)?  Same N and M, K varied around
a power-of-two-ish stride: if every workgroup's k-th block falls on the same L2 / memory channel the launch is channel-bound.
Usage on the GPU box: python tools/ubench/gemm_kstride.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, lib

dev = torch.device("cuda")
dt = torch.bfloat16
L = 4
for M, N, KS in ((1000, 1536, (6144, 6176, 6208, 6272, 6400, 5632, 4096, 4128)), (2048, 3072, (8192, 8224, 8320, 7680)),
                 (64, 3072, (8192, 8224, 8320, 7680))):
    for K in KS:
        ws = ops.linear_workspace(dt, M, K, dev)
        w0 = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]
        x = torch.randn(M, K, device=dev); y = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
        lib.ua2_debug_force_general_linear(5 if M > 64 else 0)
        args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=a, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res,
                           workspace=ws, launch=False) for a in w0]
        ops.linear_chain_timed(args, 2)
        t = ops.linear_chain_timed(args, 5)
        lib.ua2_debug_force_general_linear(0)
        print(f"M={M} N={N} K={K} stride={K // 32} KiB: {t * 1e3:7.1f} us  {2.0 * M * N * K / t / 1e9:6.1f} TFLOP/s  {t * 1e3 / K * 1e3:6.2f} ns per k", flush=True)
