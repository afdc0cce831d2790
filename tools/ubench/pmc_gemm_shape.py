#!/usr/bin/env python3
"""Minimal launch set for a PMC pass on the tiled GEMM at one shape: UA2_SHAPE=M,N,K (RESIDUAL epilogue, bf16), 4 launches.
  rocprofv3 --pmc <counters> --kernel-trace -d DIR -o NAME -- python tools/ubench/pmc_gemm_shape.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, lib
dev = torch.device("cuda")
dt = torch.bfloat16
M, N, K = (int(v) for v in os.environ.get("UA2_SHAPE", "1000,1536,6144").split(","))
w = ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt)
x = torch.randn(M, K, device=dev); y = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
ws = ops.linear_workspace(dt, M, K, dev)
lib.ua2_debug_force_general_linear(5)
torch.cuda.synchronize()
for _ in range(4):
    ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res, workspace=ws)
torch.cuda.synchronize()
print("done", float(y.abs().sum()))
