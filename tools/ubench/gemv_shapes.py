#!/usr/bin/env python3
"""Times every weight-streaming GEMV shape of the decode frame in isolation (HIP-event timed chains
over distinct per-layer weights, so each launch streams cold weights), prints us and TB/s.
Usage on the GPU box: python tools/ubench/gemv_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import (EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, PRO_NORM)

dev = torch.device("cuda")
L = int(os.environ.get("UA2_UBENCH_LAYERS", "12"))     # distinct weight sets in the chain (1 = the same weights every launch)
dt = torch.bfloat16


def packed(N, K, **kw):
    return [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt, **kw) for _ in range(L)]


def run(name, mk_args, nbytes):
    args = mk_args()
    ops.linear_chain_timed(args, 3)
    ms = ops.linear_chain_timed(args, 20)
    print(f"{name:28s} {ms*1e3:7.2f} us  {nbytes/ms/1e9:6.2f} TB/s  ({nbytes/1e6:.1f} MB)", flush=True)


for tag, C, nh, nkv, hs, I in (("trunk", 3072, 24, 8, 128, 8192), ("dec", 2048, 32, 8, 64, 8192)):
    x = torch.randn(1, C, device=dev); nw = torch.ones(C, device=dev)
    act = torch.randn(1, I, device=dev); y = torch.empty(1, max(I, (nh + 2 * nkv) * hs), device=dev)
    xr = torch.randn(1, C, device=dev)
    pos = torch.tensor([40], dtype=torch.int32, device=dev)
    cos = torch.randn(2048, hs // 2, device=dev); sin = torch.randn(2048, hs // 2, device=dev)
    q = torch.empty(1, nh * hs, device=dev)
    kp = torch.zeros(32, nkv, 64, hs, dtype=dt, device=dev); vp = torch.zeros_like(kp)
    pt = torch.arange(32, dtype=torch.int32, device=dev).view(1, 32)
    geom = ops.kv_geom(kp, vp, pt, nh, nkv, hs)
    nq = (nh + 2 * nkv) * hs
    wq = packed(nq, C, rope_head_size=hs)
    run(f"{tag} qkv {C}->{nq}", lambda: [ops.linear(dtype=dt, M=1, N=nq, K=C, w0=w, prologue=PRO_NORM, epilogue=EPI_QKV_ROPE, x=x, norm_w=nw, row_pos=pos, rope_cos=cos, rope_sin=sin, q_out=q, kv=geom, launch=False) for w in wq], nq * C * 2)
    del wq
    wo = packed(C, nh * hs)
    ya = torch.randn(1, nh * hs, device=dev)
    run(f"{tag} oproj {nh*hs}->{C}", lambda: [ops.linear(dtype=dt, M=1, N=C, K=nh * hs, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=ya, y=xr, resid=xr, launch=False) for w in wo], C * nh * hs * 2)
    del wo
    w1, w2 = packed(I, C), packed(I, C)
    run(f"{tag} swiglu {C}->2x{I}", lambda: [ops.linear(dtype=dt, M=1, N=I, K=C, w0=a_, w1=b_, prologue=PRO_NORM, epilogue=EPI_SWIGLU, x=x, norm_w=nw, y=act, launch=False) for a_, b_ in zip(w1, w2)], 2 * I * C * 2)
    del w1, w2
    wd = packed(C, I)
    run(f"{tag} down {I}->{C}", lambda: [ops.linear(dtype=dt, M=1, N=C, K=I, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=act, y=xr, resid=xr, launch=False) for w in wd], C * I * 2)
    del wd
wp = packed(2048, 3072)
h3 = torch.randn(1, 3072, device=dev); xd = torch.empty(1, 2048, device=dev)
run("projection 3072->2048", lambda: [ops.linear(dtype=dt, M=1, N=2048, K=3072, w0=w, prologue=PRO_CAST, epilogue=EPI_STORE, x=h3, y=xd, launch=False) for w in wp], 2048 * 3072 * 2)
wa = packed(12296, 2048)
pm = torch.empty(1, 769, device=dev); pi = torch.empty(1, 769, dtype=torch.int32, device=dev); lg = torch.empty(1, 12296, device=dev)
fb = torch.zeros(1, dtype=torch.int32, device=dev); nw2 = torch.ones(2048, device=dev)
run("audio_head 2048->12296", lambda: [ops.linear(dtype=dt, M=1, N=12296, K=2048, w0=w, prologue=PRO_NORM, epilogue=EPI_STORE, x=xd, norm_w=nw2, y=lg, part_max=pm, part_idx=pi, forbid=fb, launch=False) for w in wa], 12296 * 2048 * 2)
