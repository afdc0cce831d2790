#!/usr/bin/env python3
"""What a 1000-row order-free launch costs apart from its K loop: the DiT's four shapes at K = 64 ... full K (same N, same epilogue),
launch alone, rotating weight sets.  A straight line in K: intercept = boundary + ring fill + epilogue, slope = time per chunk of 32.
Usage: python tools/ubench/gemm2_fixed_cost.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_GELU, EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, GELU_TANH, PRO_CAST, ROPE_NONE, SUM_ORDER_FREE, lib

dev, dt, L = torch.device("cuda"), torch.bfloat16, 6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
os.environ["UA2_GEMM2_BMT"] = os.environ.get("UA2_GEMM2_BMT", "8")
for name, N, Kfull, epi in (("qkv (cache write)", 4608, 1536, EPI_QKV_ROPE), ("ff1 (gelu, packed)", 6144, 1536, EPI_GELU), ("gelu, fp32 y", 6144, 1536, -EPI_GELU),
                            ("store", 4608, 1536, EPI_STORE), ("store N=6144", 6144, 1536, EPI_STORE), ("o / ff2 (residual, no split)", 1536, 6144, EPI_RESIDUAL)):
    plain_y = epi < 0
    epi = abs(epi)
    row = []
    for K in (64, 384, 768, 1536, 3072, 6144):
        if K > Kfull:
            continue
        ws = [ops.pack_linear(torch.randn(N, K, device=dev) * 0.02, dt) for _ in range(L)]
        xp = torch.randn((M + 15) // 16 * 16 * K, device=dev).to(dt)
        y, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
        kw = {}
        if epi == EPI_QKV_ROPE:
            nh = N // 3 // 64
            npg = (M // 2 + 63) // 64
            kp = torch.zeros(2 * npg, nh, 64, 64, dtype=dt, device=dev); vp = torch.zeros_like(kp)
            pt = torch.arange(2 * npg, dtype=torch.int32, device=dev).view(2, npg)
            pos = torch.arange(M // 2, dtype=torch.int32, device=dev).repeat(2)[:M].contiguous()
            seq = torch.arange(2, dtype=torch.int32, device=dev).repeat_interleave(M // 2)[:M].contiguous()
            kw = dict(rope_mode=ROPE_NONE, row_pos=pos, row_seq=seq, q_out=y[:, :N // 3].contiguous(), kv=ops.kv_geom(kp, vp, pt, nh, nh, 64))
        elif epi == EPI_GELU and not plain_y:
            kw = dict(y_packed=ops.linear_workspace(dt, M, N, dev), act_kind=GELU_TANH)      # the DiT's form
        else:
            kw = dict(y=y, **(dict(act_kind=GELU_TANH) if epi == EPI_GELU else {}))
        if epi == EPI_RESIDUAL:
            kw["resid"] = res
        args = [ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=epi, x_packed=xp, sum_order=SUM_ORDER_FREE, launch=False, **kw) for w in ws]
        n0 = lib.ua2_debug_kernel_launches(b"gemm2")
        ops.linear_chain_timed(args, 2)
        assert lib.ua2_debug_kernel_launches(b"gemm2") > n0, "not on the order-free kernel"
        row.append((K, min(ops.linear_chain_timed(args, 10) for _ in range(3)) * 1e3))
    (k0, t0), (k1, t1) = row[0], row[-1]
    slope = (t1 - t0) / ((k1 - k0) / 32)
    print(f"M={M} {name:30s} " + "  ".join(f"K={k}: {t:5.1f}" for k, t in row) + f"  | per chunk {slope:.3f} us, intercept {t0 - slope * k0 / 32:.1f} us", flush=True)
