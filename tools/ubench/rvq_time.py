#!/usr/bin/env python3
"""Times ua2_rvq_encode at the live codec's sizes (one 10-s clip: 125 vectors x 6 levels x 8192 x 32; a 32-clip batch; Mimi) and
checks the codes against the C oracle.  Usage on the GPU box: python tools/ubench/rvq_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from uniaudio2_amd import ops
from oracle import rvq_oracle

for L, C, D, N in ((6, 8192, 32, 125), (1, 8192, 32, 125), (6, 8192, 32, 51), (8, 4096, 64, 51), (6, 8192, 32, 4000), (32, 2048, 256, 25)):
    g = torch.Generator().manual_seed(L * 1000 + D)
    x = torch.randn(N, D, generator=g)
    emb = torch.randn(L, C, D, generator=g) * (0.7 ** torch.arange(L).float()).view(L, 1, 1)
    xd, ed = x.cuda(), emb.cuda()
    eT = ed.transpose(1, 2).contiguous()
    codes, q = ops.rvq_encode(xd, ed, eT)
    torch.cuda.synchronize()
    ok = "n/a"
    if N <= 200:
        o_codes, o_q = rvq_oracle.rvq_encode(x.numpy(), emb.numpy())
        ok = bool((codes.cpu().numpy() == o_codes).all() and (q.cpu().numpy() == o_q).all())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ops.rvq_encode(xd, ed, eT)
    e0.record()
    for _ in range(20):
        ops.rvq_encode(xd, ed, eT)
    e1.record(); torch.cuda.synchronize()
    print(f"L={L} C={C} D={D} N={N}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call, equals oracle: {ok}", flush=True)
