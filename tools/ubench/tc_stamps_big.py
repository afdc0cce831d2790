#!/usr/bin/env python3
"""Cycle stamps of convtc_big_kernel (UA2_LIB = a -DUA2_TC_DBG=32 build): wave 0 of workgroup 0.
python tools/ubench/tc_stamps_big.py C dil T"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uniaudio2_amd import _lib, ops

C, dil, T = (int(v) for v in sys.argv[1:4])
g = torch.Generator().manual_seed(0)
x = ops.tc_pack(torch.randn(1, C, T, generator=g).cuda())
w = (torch.randn(C, C, 7, generator=g) / (C * 7) ** 0.5).cuda()
hi, lo = ops.pack_conv_weight_x3(w)
w2 = (torch.randn(C, C, 1, generator=g) / C ** 0.5).cuda()
kw = dict(dilation=dil, pad_left=dil * 6, Tout=T, bias=torch.randn(C).cuda(), post_act=1, post_alpha=torch.tensor([0.2]).cuda(), variant=3,
          fused2=(*ops.pack_conv_weight_x3(ops.tc_w2_order(w2)), torch.randn(C).cuda(), torch.tensor([0.3]).cuda()))
for _ in range(3):
    ops.conv1d_tc(x, hi, lo, 7, C, **kw)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
assert raw.ua2_tc_stamps(buf) == 0
v = list(buf)
nch = 7 * (C // 32)
print(f"big kernel C={C} dil={dil} T={T}: prologue issue {v[1] - v[0]}, drain {v[2] - v[1]}, barrier {v[3] - v[2]}")
prev = v[3]
for c in range(nch):
    if c == 7:
        print(f"  group boundary: wait {v[21] - v[20]}, barrier {v[22] - v[21]}")
        prev = v[22]
    print(f"  chunk {c}: {v[4 + c] - prev}")
    prev = v[4 + c]
ntt = 4 if C == 64 else 8
print(f"  epilogue: w2 request {v[25] - v[24]}; per time tile " + ", ".join(str(v[25 + i + 1] - v[25 + i]) for i in range(ntt - 1)) + f"; last tile + stores issued {v[40] - v[25 + ntt - 1]}; store drain {v[41] - v[40]}")
print(f"  total {v[41] - v[0]} cycles")
