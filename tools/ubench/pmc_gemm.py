#!/usr/bin/env python3
"""Minimal launch set for a PMC pass on the 128-row tiled GEMM (gemm_kernel<1, 2>: RMSNorm + fc_1/fc_2 + SwiGLU at
M = 6272 rows, 3072 -> 2 x 8192, bf16 — the prefill shape of SURVEY.md §8d config 3): 4 launches, no graphs.  Run under
  rocprofv3 --pmc <counters> --kernel-trace -d DIR -o NAME -- python tools/ubench/pmc_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_SWIGLU, PRO_NORM, lib
dev = torch.device("cuda")
dt = torch.bfloat16
M, C, I = 6272, 3072, 8192
w1 = ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt)
w2 = ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt)
x = torch.randn(M, C, device=dev); nw = torch.ones(C, device=dev); y = torch.empty(M, I, device=dev)
ws = ops.linear_workspace(dt, M, C, dev)
lib.ua2_debug_force_general_linear(5)
torch.cuda.synchronize()
for _ in range(4):
    ops.linear(dtype=dt, M=M, N=I, K=C, w0=w1, w1=w2, prologue=PRO_NORM, epilogue=EPI_SWIGLU, x=x, norm_w=nw, y=y, workspace=ws)
torch.cuda.synchronize()
print("done", float(y.abs().sum()))
