#!/usr/bin/env python3
"""One ua2_conv1d_tc shape, event-bracketed median of 30 launches (for knock-out builds: UA2_LIB=...).  python tools/ubench/tc_one.py C K dil T [res]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uniaudio2_amd import ops

Cc, K, d, T = (int(v) for v in sys.argv[1:5])
res = len(sys.argv) > 5 and sys.argv[5] == "res"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = ops.tc_pack(torch.randn(1, Cc, T, device=dev))
r = ops.tc_pack(torch.randn(1, Cc, T, device=dev))
hi, lo = ops.pack_conv_weight_x3(torch.randn(Cc, Cc, K, device=dev) / (Cc * K) ** 0.5)
bias, alpha = torch.randn(Cc, device=dev), torch.full((1,), 0.25, device=dev)
call = lambda: ops.conv1d_tc(x, hi, lo, K, Cc, dilation=d, pad_left=(K - 1) * d // 2, Tout=T, bias=bias, post_act=1, post_alpha=alpha, residual=r if res else None)
for _ in range(3):
    call()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record()
    ts.append((e0, e1))
torch.cuda.synchronize()
v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
print(f"C={Cc} K={K} d={d} T={T} {'res' if res else ''} lib={os.path.basename(os.environ.get('UA2_LIB', 'product'))}: {v[len(v) // 2]:.1f} us", flush=True)
