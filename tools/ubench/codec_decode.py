#!/usr/bin/env python3
"""ScalarModel.decode of one 20-s window (placeholder widths of bench.py): ms per decode, and per-layer times of the decode
chain (one HIP-event pair around every launch of one pass).  python tools/ubench/codec_decode.py [--layers]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from uniaudio2_amd import ops
from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel

dev = torch.device("cuda:0")
torch.manual_seed(1)
sq = ScalarModel(**bench.SCALAR_CFG).to(dev).prepare()
lat = torch.tanh(torch.randn(1, 136, 500, device=dev))
for _ in range(3):
    wav = sq.decode(lat)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    wav = sq.decode(lat)
e1.record(); torch.cuda.synchronize()
print(f"decode: {e0.elapsed_time(e1) / 10:.3f} ms per 20-s window, tc path = {sq._dec_tc}, graph = {bool(sq._graphs)}")
e0.record()
for _ in range(10):
    wav2 = sq.decode(lat, use_graph=False)
e1.record(); torch.cuda.synchronize()
print(f"decode, launches issued one by one: {e0.elapsed_time(e1) / 10:.3f} ms; identical: {torch.equal(wav, wav2)}")
if "--layers" in sys.argv:
    rows = []
    for name in ("conv1d", "conv1d_tc", "tc_pack"):
        real = getattr(ops, name)

        def wrap(*a, _real=real, _name=name, **kw):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            y = _real(*a, **kw)
            s1.record()
            x = a[0]
            rows.append((_name, tuple(x.shape), kw.get("dilation", 1), "fused" if kw.get("fused2") else ("res" if kw.get("residual") is not None else ""),
                         kw.get("out_phases", 1), s0, s1))
            return y
        setattr(ops, name, wrap)
    for _ in range(2):
        rows.clear()
        sq.decode(lat, use_graph=False)
    torch.cuda.synchronize()
    tot = 0.0
    agg = {}
    for name, shp, dil, kind, ph, s0, s1 in rows:
        us = s0.elapsed_time(s1) * 1e3
        tot += us
        if "--compact" in sys.argv:
            agg.setdefault((name, shp, kind, ph), []).append(us)
        else:
            print(f"{name:10s} in={shp} dil={dil} phases={ph} {kind:5s} {us:7.1f} us")
    for (name, shp, kind, ph), v in agg.items():
        print(f"{name:10s} in={shp} phases={ph} {kind:5s} n={len(v)} mean {sum(v) / len(v):7.1f} us")
    print(f"sum of launches (event-bracketed, includes launch gaps): {tot:.1f} us")
