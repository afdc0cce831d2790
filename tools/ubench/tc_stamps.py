#!/usr/bin/env python3
"""Cycle stamps of the pipelined tc conv kernel (UA2_LIB = a -DUA2_TC_DBG=32 build): one launch of a given shape, then the
phase table of wave 0 / workgroup 0 / second tile.  python tools/ubench/tc_stamps.py C K dil T [fused|res] [phases]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uniaudio2_amd import _lib, ops

Cin, Cout, K, dil, T = (int(v) for v in sys.argv[1:6])
kind = sys.argv[6] if len(sys.argv) > 6 else ""
phases = int(sys.argv[7]) if len(sys.argv) > 7 else 1
g = torch.Generator().manual_seed(0)
x = ops.tc_pack(torch.randn(1, Cin, T, generator=g).cuda())
w = (torch.randn(Cout * phases, Cin, K, generator=g) / (Cin * K) ** 0.5).cuda()
hi, lo = ops.pack_conv_weight_x3(w)
kw = dict(dilation=dil, pad_left=dil * (K - 1), Tout=T * phases, bias=torch.randn(Cout).cuda(), post_act=1, post_alpha=torch.tensor([0.2]).cuda(),
          out_phases=phases, variant=2)
if kind == "fused":
    w2 = (torch.randn(Cout, Cout, 1, generator=g) / Cout ** 0.5).cuda()
    kw["fused2"] = (*ops.pack_conv_weight_x3(w2), torch.randn(Cout).cuda(), torch.tensor([0.3]).cuda())
elif kind == "res":
    kw["residual"] = ops.tc_pack(torch.randn(1, Cout, T, generator=g).cuda())
for _ in range(3):
    ops.conv1d_tc(x, hi, lo, K, Cout, **kw)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
assert raw.ua2_tc_stamps(buf) == 0
v = list(buf)
upt = Cin // 32 // max(1, {7: 1, 2: 4 if (Cin // 32) % 4 == 0 else 2, 1: 4}[K])
print(f"shape Cin={Cin} Cout={Cout} K={K} dil={dil} T={T} {kind} phases={phases}: units per tile {upt}")
names = ["chunk loop", "wait DMA", "barrier", "DMA issue"]
prev_end = None
for ug in range(min(upt, 5)):
    s = v[ug * 8: ug * 8 + 5]
    if prev_end is not None:
        print(f"  (gap before unit {ug}: {s[0] - prev_end})")
    print(f"  unit {ug}: " + ", ".join(f"{n} {s[i + 1] - s[i]}" for i, n in enumerate(names)))
    prev_end = s[4]
e = v[40:45]
last_unit_end = v[(min(upt, 5) - 1) * 8 + 4] if upt <= 5 else None
if kind == "fused":
    print(f"  epilogue: h image {e[1] - e[0]}, barrier {e[2] - e[1]}, 1x1 MFMAs {e[3] - e[2]}, residual + split + stores {e[4] - e[3]}")
else:
    print(f"  epilogue: {e[4] - e[0]}")
print(f"  whole tile (unit 0 start -> after epilogue): {e[4] - v[0]} cycles (100 MHz ticks if s_memtime is the realtime counter)")
