#!/usr/bin/env python3
"""Minimal launch set for a PMC pass on the dominant kernel (gemv_kernel<1,1,2,4,true>, 3072 -> 2x8192 SwiGLU GEMV):
8 launches over 8 distinct weight sets (HBM-cold), no graphs.  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o NAME -- python tools/ubench/pmc_swiglu.py
(counters in their own pass, MI355X_MICROARCH.md §HBM; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_SWIGLU, PRO_NORM
dev = torch.device("cuda")
dt = torch.bfloat16
C, I, L = 3072, 8192, 8
w1 = [ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt) for _ in range(L)]
w2 = [ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt) for _ in range(L)]
x = torch.randn(1, C, device=dev); nw = torch.ones(C, device=dev); y = torch.empty(1, I, device=dev)
torch.cuda.synchronize()
for l in range(L):
    ops.linear(dtype=dt, M=1, N=I, K=C, w0=w1[l], w1=w2[l], prologue=PRO_NORM, epilogue=EPI_SWIGLU, x=x, norm_w=nw, y=y)
torch.cuda.synchronize()
print("done", float(y.abs().sum()))
