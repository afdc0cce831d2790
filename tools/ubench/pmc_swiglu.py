#!/usr/bin/env python3
"""Minimal launch set for a PMC pass on the dominant kernel (gemv_kernel<1,4,2,4,true>: scaled-RMSNorm + fc_1/fc_2 + SwiGLU GEMV,
3072 -> 2x8192, the form the B = 1 frame runs since round 3): 8 launches over 8 distinct weight sets (HBM-cold), no graphs.
Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o NAME -- python tools/ubench/pmc_swiglu.py
(counters in their own pass, MI355X_MICROARCH.md §HBM; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_SWIGLU
PRO_SCALED = 4
dev = torch.device("cuda")
dt = torch.bfloat16
C, I, L = 3072, 8192, 8
w1 = [ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt) for _ in range(L)]
w2 = [ops.pack_linear(torch.randn(I, C, device=dev) * 0.02, dt) for _ in range(L)]
xh = torch.randn(1, C, device=dev).to(dt); ssq = torch.rand(1, C // 16, device=dev) * 16; y = torch.empty(1, I, device=dev)
torch.cuda.synchronize()
for l in range(L):
    ops.linear(dtype=dt, M=1, N=I, K=C, w0=w1[l], w1=w2[l], prologue=PRO_SCALED, epilogue=EPI_SWIGLU, x_h=xh, x_ssq=ssq, eps=1e-5, y=y)
torch.cuda.synchronize()
print("done", float(y.abs().sum()))
