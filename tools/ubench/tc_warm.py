#!/usr/bin/env python3
"""Is a wide decode conv bound by where its FILTER comes from?  One ua2_conv1d_tc launch, event-bracketed, (a) repeated back to back (the
filter stays in the L2 of the XCDs that read it), (b) with 96 MB streamed through the caches in front of every launch (filter from HBM),
(c) as (b) but the filter touched by a small read kernel right in front of the launch (what a cross-launch prefetch would give).
python tools/ubench/tc_warm.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uniaudio2_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
SHAPES = [("k7 d1", 512, 1500, 7, 1, False), ("k7 d9", 512, 1500, 7, 9, False), ("k1 res", 512, 1500, 1, 1, True),
          ("k7 d1", 256, 7500, 7, 1, False), ("k7 d9", 256, 7500, 7, 9, False), ("k1 res", 256, 7500, 1, 1, True)]
junk = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
junk2 = torch.empty_like(junk)


def bracket(fn, pre=None, n=30):
    ts = []
    for _ in range(n):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    return v[len(v) // 2]


for name, Cc, T, K, d, res in SHAPES:
    x = ops.tc_pack(torch.randn(1, Cc, T, device=dev))
    w = torch.randn(Cc, Cc, K, device=dev) / (Cc * K) ** 0.5
    hi, lo = ops.pack_conv_weight_x3(w)
    bias = torch.randn(Cc, device=dev)
    alpha = torch.full((1,), 0.25, device=dev)
    call = lambda: ops.conv1d_tc(x, hi, lo, K, Cc, dilation=d, pad_left=(K - 1) * d // 2, Tout=T, bias=bias, post_act=1, post_alpha=alpha,
                                 residual=x if res else None)
    for _ in range(3):
        call()
    thrash = lambda: junk2.copy_(junk)
    def thrash_touch():
        junk2.copy_(junk)
        hi.view(torch.int32).sum(); lo.view(torch.int32).sum()
    a = bracket(call)
    b = bracket(call, thrash)
    c = bracket(call, thrash_touch)
    print(f"C={Cc:4d} T={T:5d} {name:7s} filter {hi.numel() * hi.element_size() * 2 / 1e6:5.2f} MB | back to back {a:6.1f} us | caches streamed over {b:6.1f} us | "
          f"... then the filter touched {c:6.1f} us", flush=True)
