#!/usr/bin/env python3
"""Minimal launch set for a PMC pass on the order-free GEMM (gemm2_kernel<SWIGLU, 16, 4, 2>: fc_1/fc_2 + SwiGLU at M = 6272 rows,
3072 -> 2 x 8192, bf16 — the prefill shape of SURVEY.md §8d config 3; operand already packed): 4 launches, no graphs.  Run under
  rocprofv3 --pmc <counters> --kernel-trace -d DIR -o NAME -- python tools/ubench/pmc_gemm2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_SWIGLU, PRO_CAST, SUM_ORDER_FREE
dev, dt = torch.device("cuda"), torch.bfloat16
M, C, I = 6272, 3072, 8192
w1 = ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt)
w2 = ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt)
xp = torch.randn(M * C, device=dev).to(dt)
y = torch.empty(M, I, device=dev)
torch.cuda.synchronize()
for _ in range(4):
    ops.linear(dtype=dt, M=M, N=I, K=C, w0=w1, w1=w2, prologue=PRO_CAST, epilogue=EPI_SWIGLU, x_packed=xp, y=y, sum_order=SUM_ORDER_FREE)
torch.cuda.synchronize()
print("done", float(y.abs().sum()))
