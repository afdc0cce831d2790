"""Tile sweep for the bf16 x 3 conv (ua2_conv1d precision 1) on the decoder's layer shapes: one hipGraph of N launches per
(shape, tile) pair, timed with events.  The tile comes from the UA2_CONV_PIPE experiment hook of the launcher; "!" marks bits that differ from the plain kernel's.
Usage: python tools/ubench/conv_shapes.py N [tile ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uniaudio2_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)

SHAPES = [  # (name, C, T, K, dilation)
    ("L0 k7 d1", 512, 1500, 7, 1), ("L0 k7 d9", 512, 1500, 7, 9), ("L0 k1", 512, 1500, 1, 1),
    ("L1 k7 d1", 256, 7500, 7, 1), ("L1 k7 d9", 256, 7500, 7, 9), ("L1 k1", 256, 7500, 1, 1),
    ("L2 k7 d9", 128, 30000, 7, 9), ("L3 k7 d9", 64, 120000, 7, 9), ("L4 k7 d9", 32, 240000, 7, 9),
]
# UA2_CONV_PIPE = "ntt,rpw,rt,tpw,gpu" (0 = automatic) for the pipelined kernel, "off" = the plain kernel (the reference bits)
TILES = ["off", None] + sys.argv[2:]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * N)


FUSED = os.environ.get("UA2_UBENCH_FUSED", "0") == "1"      # the residual-unit launches (conv + 1 x 1 conv fused) of the big-T levels
if FUSED:
    SHAPES = [("F2 k7 d9", 128, 30000, 7, 9), ("F3 k7 d1", 64, 120000, 7, 1), ("F3 k7 d9", 64, 120000, 7, 9), ("F4 k7 d1", 32, 240000, 7, 1), ("F4 k7 d9", 32, 240000, 7, 9)]
for name, Cc, T, K, d in SHAPES:
    x = torch.randn(1, Cc, T, device=dev)
    w = torch.randn(Cc, Cc, K, device=dev) / (Cc * K) ** 0.5
    hi, lo = ops.pack_conv_weight_x3(w)
    bias = torch.randn(Cc, device=dev)
    alpha = torch.full((1,), 0.25, device=dev)
    pad = (K - 1) * d // 2
    if FUSED:
        w2h, w2l = ops.pack_conv_weight_x3(torch.randn(Cc, Cc, 1, device=dev) / Cc ** 0.5)
        f2 = (w2h, w2l, bias, alpha)
    ref = None
    row = []
    for tile in TILES:
        if tile is None:
            os.environ.pop("UA2_CONV_PIPE", None)
        else:
            os.environ["UA2_CONV_PIPE"] = tile
        if FUSED:
            call = lambda: ops.conv1d(x, hi, K, Cc, dilation=d, pad_left=(K - 1) * d, Tout=T, bias=bias, pre_act=0, post_act=1, post_alpha=alpha,
                                      residual=x, w_lo=lo, fused2=f2)
        else:
            call = lambda: ops.conv1d(x, hi, K, Cc, dilation=d, pad_left=pad, Tout=T, bias=bias, pre_act=1, pre_alpha=alpha, residual=x, w_lo=lo)
        try:
            y = call()
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            ok = torch.equal(y, ref)
            us = timed(call)
            row.append(f"{tile or 'auto'}:{us:.1f}{'' if ok else '!'}")
        except Exception as ex:  # noqa: BLE001 — sweep: a tile the launcher refuses is just skipped
            row.append(f"{tile}:ERR")
    print(f"{name:10s} C={Cc:4d} T={T:6d}  " + "  ".join(row), flush=True)
