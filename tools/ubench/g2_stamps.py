#!/usr/bin/env python3
"""Cycle stamps of the order-free GEMM's phase loop (UA2_LIB = a -DUA2_G2_DBG=8 build): one launch, then per chunk 8 .. 11 the
phase table of wave 0 (group 0) and wave 4 (group 1) of workgroup 0.   python tools/ubench/g2_stamps.py M N K [bmt]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import _lib, ops
from uniaudio2_amd._lib import EPI_STORE, PRO_CAST, SUM_ORDER_FREE
M, N, K = (int(v) for v in sys.argv[1:4])
if len(sys.argv) > 4:
    os.environ["UA2_GEMM2_BMT"] = sys.argv[4]
dev, dt = torch.device("cuda"), torch.bfloat16
w = ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt)
xp = torch.randn((M + 15) // 16 * 16 * K, device=dev).to(dt)
y = torch.empty(M, N, device=dev)
for _ in range(3):
    ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_STORE, x_packed=xp, y=y, sum_order=SUM_ORDER_FREE)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
assert raw.ua2_g2_stamps(buf) == 0
v = list(buf)
names = ["reads issued", "DMA issued", "vm+lgkm wait", "barrier (L end)", "MFMAs", "vmcnt wait", "barrier (C end)"]
print(f"M={M} N={N} K={K} bmt={os.environ.get('UA2_GEMM2_BMT', 'auto')}: cycles per segment, chunks 8..11")
for g in range(2):
    print(f" group {g}:")
    for c in range(4):
        s = v[(g * 4 + c) * 8:(g * 4 + c) * 8 + 8]
        seg = [s[i + 1] - s[i] for i in range(7)]
        nxt = v[(g * 4 + c + 1) * 8] - s[7] if c < 3 else 0
        print("   chunk %2d: " % (8 + c) + "  ".join(f"{n} {d}" for n, d in zip(names, seg)) + f"  | L {s[4]-s[0]} C {s[7]-s[4]} total {s[7]-s[0]} (+{nxt} to next)")
# relative timing of the two groups: start of group 1's L(c) against group 0's C(c) start
for c in range(4):
    a, b = v[(0 * 4 + c) * 8 + 4], v[(1 * 4 + c) * 8 + 0]
    print(f" chunk {8 + c}: group 1 enters L {b - a:+d} cycles after group 0 enters C")
