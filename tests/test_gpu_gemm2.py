"""The order-free many-row GEMM (csrc/ua2_gemm2.hip, ua2hip.h `sum_order = UA2_SUM_ORDER_FREE`) against the row-invariant
kernels of ua2_gemm.hip on the same launches.

Contract under test: same function, same per-value epilogue operations, another order of the K sum (one MFMA chain per K slab
instead of the decode kernel's wave ranges) — so fp32 outputs agree to fp32 summation noise (bounded here by 2e-5 sqrt(K) on
unit-scale sums, the bar of the K-split test), bf16 outputs (packed operand hand-offs, K/V pages) to one bf16 rounding of that
noise, and the launch is deterministic (same bits every time: the race screen of the two-group phase structure).  The invariant
kernels themselves are pinned on the decode kernel, which is pinned on the oracle (tests/test_gpu_invariance.py,
tests/test_gpu_lm.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, K, what):
    if a.dtype == torch.bfloat16:
        a, b = a.float(), b.float()
        tol = 2e-5 * K ** 0.5 + 2.0 ** -7 * b.abs()            # the fp32 noise can flip one bf16 rounding
        bad = (a - b).abs() > tol
        assert not bad.any(), (what, int(bad.sum()), float((a - b).abs().max()))
        assert (a != b).float().mean().item() < 0.02, (what, "more than 2 % of the bf16 values differ")
    else:
        err = (a - b).abs().max().item()
        assert err < 2e-5 * K ** 0.5, (what, err)


@pytest.fixture
def g2env():
    keys = ("UA2_GEMM2_BMT", "UA2_GEMM2_OFF", "UA2_GEMM2_MIN_ROWS", "UA2_GEMM2_FORCE", "UA2_GEMM_NO_KSPLIT", "UA2_GEMM2_NO_TAIL", "UA2_GEMM2_R5_FORMS")
    saved = {k: os.environ.get(k) for k in keys}

    def set_(**kw):
        for k in keys:
            os.environ.pop(k, None)
        for k, v in kw.items():
            os.environ[k] = str(v)
        _refresh()
    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    _refresh()


def _refresh():
    """The launchers read their UA2_* knobs once (ua2hip.h ua2_debug_refresh_env): tell them the environment changed."""
    from uniaudio2_amd._lib import lib
    lib.ua2_debug_refresh_env()


@pytest.mark.parametrize("bmt", [16, 8])
def test_every_epilogue_agrees_with_the_invariant_kernels(bmt, g2env):
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import (EPI_GELU, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, GATE_SIGMOID_SECOND, GELU_TANH, NORM_LAYERNORM, PRO_CAST,
                                    PRO_NORM, SUM_ORDER_FREE)
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(5 + bmt)
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, generator=g)).to(dev)
    for (N, K) in ((1536, 1056), (512, 3072), (4608, 1536)):
        p0, p1 = ops.pack_linear(mk(N, K, s=K ** -0.5), dt), ops.pack_linear(mk(N, K, s=K ** -0.5), dt)
        bias, bias1, gate, nw, nb = mk(N, s=0.3), mk(N, s=0.3), mk(N, s=0.5), 1.0 + mk(K, s=0.1), mk(K, s=0.1)
        Mmax = 1300
        x, res = mk(Mmax, K), mk(Mmax, N, s=2.0)

        def run(M, epi, order, **opt):
            outs = {}
            kw = dict(dtype=dt, M=M, N=N, K=K, w0=p0, epilogue=epi, workspace=ops.linear_workspace(dt, M, K, dev), sum_order=order)
            if opt.get("ln"):
                kw.update(prologue=PRO_NORM, x=x[:M].contiguous(), norm_w=nw, norm_b=nb, eps=1e-6, norm_kind=NORM_LAYERNORM)
            elif opt.get("cast"):
                kw.update(prologue=PRO_CAST, x=x[:M].contiguous())
            else:
                kw.update(prologue=PRO_NORM, x=x[:M].contiguous(), norm_w=nw, eps=1e-5)
            if opt.get("bias"):
                kw.update(bias=bias, bias1=bias1 if epi == EPI_SWIGLU else None)
            if epi == EPI_SWIGLU:
                kw.update(w1=p1, act_kind=opt.get("act", 0))
            if epi == EPI_GELU:
                kw.update(act_kind=opt.get("act", 0))
            if epi == EPI_RESIDUAL:
                kw.update(resid=res[:M].contiguous(), out_scale=gate if opt.get("gate") else None)
            if opt.get("y", True):
                outs["y"] = torch.zeros(M, N, device=dev)
                kw.update(y=outs["y"])
            if opt.get("packed"):
                outs["packed"] = torch.zeros((M + 15) // 16 * 16 * N, dtype=dt, device=dev)
                kw.update(y_packed=outs["packed"])
            ops.linear(**kw)
            torch.cuda.synchronize()
            return outs

        cases = [(EPI_STORE, dict()), (EPI_STORE, dict(bias=True, cast=True)),
                 (EPI_RESIDUAL, dict(cast=True)), (EPI_RESIDUAL, dict(bias=True, gate=True, ln=True)),
                 (EPI_SWIGLU, dict()), (EPI_SWIGLU, dict(bias=True, act=GATE_SIGMOID_SECOND, packed=True)), (EPI_SWIGLU, dict(packed=True, y=False)),
                 (EPI_GELU, dict(bias=True)), (EPI_GELU, dict(act=GELU_TANH, packed=True, y=False, ln=True))]
        for epi, opt in cases:
            for M in (1300, 1000, 257):
                g2env(UA2_GEMM2_OFF=1)
                want = run(M, epi, SUM_ORDER_FREE, **opt)              # the switch-off hook: the invariant kernels
                g2env()
                assert all(torch.equal(want[k], v) for k, v in run(M, epi, 0, **opt).items())   # ... which the default contract always takes
                g2env(UA2_GEMM2_BMT=bmt)
                got, again = run(M, epi, SUM_ORDER_FREE, **opt), run(M, epi, SUM_ORDER_FREE, **opt)
                assert want.keys() == got.keys() and len(want) > 0
                differs = False
                for k in want:
                    assert torch.equal(got[k], again[k]), (N, K, epi, opt, M, k, "not deterministic")
                    _close(got[k], want[k], K, (N, K, epi, opt, M, k))
                    differs |= not torch.equal(got[k], want[k])
                    assert got[k].float().abs().sum().item() > 0
                if K >= 1536:
                    assert differs, (N, K, epi, opt, M, "bit-identical to the invariant kernel: the order-free kernel did not run")


@pytest.mark.parametrize("bmt", [16, 8])
@pytest.mark.parametrize("nh,nkv,hs,C,M", [(24, 8, 128, 3072, 700), (24, 24, 64, 1536, 1000), (32, 8, 64, 2048, 700)])
def test_qkv_epilogues_agree(nh, nkv, hs, C, M, bmt, g2env):
    """Fused norm + q|k|v + (RoPE) + paged K/V append: the LM's forms (half-split rotation: head size 128 in the trunk, 64 in the
    depth decoder) and the DiT's (no rotation, bias, head size 64).  Rows of three sequences at scattered positions."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_QKV_ROPE, PRO_NORM, ROPE_NONE, SUM_ORDER_FREE
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(hs + bmt)
    nq = (nh + 2 * nkv) * hs
    dit = nh == nkv
    extra = dict(rope_mode=ROPE_NONE, bias=(0.1 * torch.randn(nq, generator=g)).to(dev)) if dit else {}
    w = ops.pack_linear((torch.randn(nq, C, generator=g) * C ** -0.5).to(dev), dt, **({} if dit else dict(rope_head_size=hs)))
    x = torch.randn(M, C, generator=g).to(dev)
    nw = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    nseq = 3
    per = (M + nseq - 1) // nseq
    pos = torch.cat([torch.arange(per) for _ in range(nseq)])[:M].to(torch.int32).to(dev)
    seq = torch.arange(nseq).repeat_interleave(per)[:M].to(torch.int32).to(dev)
    npg = (per + 63) // 64
    ang = torch.rand(2048, hs // 2, generator=g) * 6.28
    cos, sin = ang.cos().to(dev), ang.sin().to(dev)
    pt = torch.randperm(nseq * npg, generator=g).to(torch.int32).view(nseq, npg).to(dev)
    outs = []
    for order, env in ((0, {}), (SUM_ORDER_FREE, dict(UA2_GEMM2_BMT=bmt)), (SUM_ORDER_FREE, dict(UA2_GEMM2_BMT=bmt))):
        g2env(**env)
        kp = torch.zeros(nseq * npg, nkv, 64, hs, dtype=dt, device=dev)
        vp = torch.zeros_like(kp)
        q = torch.zeros(M, nh * hs, device=dev)
        ops.linear(dtype=dt, M=M, N=nq, K=C, w0=w, prologue=PRO_NORM, epilogue=EPI_QKV_ROPE, x=x, norm_w=nw, row_pos=pos, row_seq=seq,
                   rope_cos=cos, rope_sin=sin, q_out=q, kv=ops.kv_geom(kp, vp, pt, nh, nkv, hs), workspace=ops.linear_workspace(dt, M, C, dev),
                   sum_order=order, **extra)
        torch.cuda.synchronize()
        outs.append((q, kp, vp))
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b)
    for name, a, b in zip(("q", "k", "v"), outs[1], outs[0]):
        _close(a, b, C, (name, nh, hs, bmt))
    assert not torch.equal(outs[1][0], outs[0][0])
    assert outs[1][1].float().abs().sum() > 0 and outs[1][2].float().abs().sum() > 0


def test_k_slabs_on_small_grids(g2env):
    """A long-K RESIDUAL launch on a grid too small for the device runs as K slabs + the fixed-order combine (split_ws)."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, SUM_ORDER_FREE
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(9)
    M, N, K = 1000, 1536, 6144
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt)
    x, res = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    bias, gate = (0.3 * torch.randn(N, generator=g)).to(dev), (0.5 * torch.randn(N, generator=g)).to(dev)

    def run(order, split_elems, **env):
        g2env(**env)
        y = torch.zeros(M, N, device=dev)
        sw = torch.full((split_elems,), float("nan"), device=dev) if split_elems else None
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res, bias=bias, out_scale=gate,
                   workspace=ops.linear_workspace(dt, M, K, dev), split_ws=sw, sum_order=order)
        torch.cuda.synchronize()
        return y

    base = run(0, 0)
    # the tile-choice rule: 48 tiles of 128 x 256 are too few for the order-free kernel — without scratch for slabs the launch goes
    # back to the invariant kernels (same bits as the default contract); with four slabs (192 workgroups) it takes it
    assert torch.equal(run(SUM_ORDER_FREE, 0), base)
    four, again = run(SUM_ORDER_FREE, 4 * M * N), run(SUM_ORDER_FREE, 4 * M * N)
    assert torch.equal(four, again) and not torch.isnan(four).any()
    assert 0 < (four - base).abs().max().item() < 2e-5 * K ** 0.5
    # the kernel itself on this shape, tile forced: one chain, two slabs, four slabs, slabs switched off
    one = run(SUM_ORDER_FREE, 0, UA2_GEMM2_BMT=8)
    for y in (one, run(SUM_ORDER_FREE, 4 * M * N, UA2_GEMM2_BMT=8), run(SUM_ORDER_FREE, 2 * M * N + 3, UA2_GEMM2_BMT=8)):
        assert 0 < (y - base).abs().max().item() < 2e-5 * K ** 0.5
    assert torch.equal(run(SUM_ORDER_FREE, 4 * M * N, UA2_GEMM2_BMT=8, UA2_GEMM_NO_KSPLIT=1), one)
    assert torch.equal(run(SUM_ORDER_FREE, 4 * M * N, UA2_GEMM2_BMT=8), four) and not torch.equal(one, four)


@pytest.mark.parametrize("M,N,K,bmt", [(6272, 1024, 3072, 16), (8000, 1536, 1536, 16), (1000, 1536, 6144, 8), (333, 512, 1056, 8), (2048, 3072, 8192, 16)])
def test_phase_structure_is_repeatable(M, N, K, bmt, g2env):
    """Race screen for the two-group phase loop (hand-counted vmcnt, raw s_barrier, LDS-DMA into slots another group has just
    read): 25 launches on a busy device give the same bits, with a second stream streaming memory beside them, and those bits
    agree with the invariant kernel's to summation noise in EVERY element (a stale or early fragment is an O(1) error)."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, SUM_ORDER_FREE
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt)
    x, res = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    ws = ops.linear_workspace(dt, M, K, dev)

    def run(order):
        y = torch.zeros(M, N, device=dev)
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res, workspace=ws, sum_order=order)
        return y

    g2env()
    base = run(0)
    g2env(UA2_GEMM2_BMT=bmt)
    first = run(SUM_ORDER_FREE)
    torch.cuda.synchronize()
    assert (first - base).abs().max().item() < 2e-5 * K ** 0.5
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=dev)
    for i in range(25):
        with torch.cuda.stream(side):
            junk.add_(1.0)
        y = run(SUM_ORDER_FREE)
        torch.cuda.synchronize()
        assert torch.equal(y, first), (i, (y - first).abs().max().item())


@pytest.mark.parametrize("bmt", [16, 8])
def test_scaled_norm_handover_on_both_sides(bmt, g2env):
    """Round 6: the scaled-norm hand-over (ua2hip.h y_norm_w / UA2_PRO_SCALED; lit_model.py:883-890 folded around :424 / :591) in the
    order-free kernel.  Producer: a RESIDUAL (and a STORE) launch emits RNE_bf16(y * w_next) in fragment order + per-16-column sums of
    squares.  Consumer: SWIGLU / STORE / q|k|v-RoPE launches read that operand and scale by rstd[m] (reduced by the small launch in
    front).  Against the invariant kernels' hand-over: y and the sums of squares to fp32 summation noise, the bf16 operand to a
    rounding flip, the consumers' outputs to the order-free tolerance — and EXACTLY when the consumer is fed the same hand-over."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, SUM_ORDER_FREE, lib
    PRO_SCALED = 4
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(70 + bmt)
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, generator=g)).to(dev)
    M, C, Kp, I = 900, 3072, 2048, 4096
    wp = ops.pack_linear(mk(C, Kp, s=Kp ** -0.5), dt)                       # producer: Kp -> C
    w0, w1 = ops.pack_linear(mk(I, C, s=C ** -0.5), dt), ops.pack_linear(mk(I, C, s=C ** -0.5), dt)
    nh, nkv, hs = 8, 4, 128
    nq = (nh + 2 * nkv) * hs
    wq = ops.pack_linear(mk(nq, C, s=C ** -0.5), dt, rope_head_size=hs)
    x, res, nw = mk(M, Kp), mk(M, C, s=2.0), 1.0 + mk(C, s=0.1)
    pos = torch.arange(M, dtype=torch.int32, device=dev)
    seq = torch.zeros(M, dtype=torch.int32, device=dev)
    npg = (M + 63) // 64
    pt = torch.randperm(npg, generator=g).to(torch.int32).view(1, npg).to(dev)
    ang = torch.rand(1024, hs // 2, generator=g) * 6.28
    cos, sin = ang.cos().to(dev), ang.sin().to(dev)

    def produce(order, epi):
        y = torch.zeros(M, C, device=dev)
        pk = ops.linear_workspace(dt, M, C, dev).zero_()
        ssq = torch.zeros(M, C // 16, device=dev)
        ops.linear(dtype=dt, M=M, N=C, K=Kp, w0=wp, prologue=PRO_CAST, epilogue=epi, x=x, y=y, resid=res if epi == EPI_RESIDUAL else None,
                   y_norm_w=nw, y_packed=pk, y_ssq=ssq, workspace=ops.linear_workspace(dt, M, Kp, dev), sum_order=order)
        torch.cuda.synchronize()
        return y, pk, ssq

    def consume(order, pk, ssq):
        outs = {}
        ws = ops.linear_workspace(dt, M, C, dev)
        outs["glu"] = torch.zeros(M, I, device=dev)
        ops.linear(dtype=dt, M=M, N=I, K=C, w0=w0, w1=w1, prologue=PRO_SCALED, epilogue=EPI_SWIGLU, x_packed=pk, x_ssq=ssq, eps=1e-5, y=outs["glu"],
                   workspace=ws, sum_order=order)
        outs["store"] = torch.zeros(M, I, device=dev)
        ops.linear(dtype=dt, M=M, N=I, K=C, w0=w0, prologue=PRO_SCALED, epilogue=EPI_STORE, x_packed=pk, x_ssq=ssq, eps=1e-5, y=outs["store"],
                   workspace=ws, sum_order=order)
        kp = torch.zeros(npg, nkv, 64, hs, dtype=dt, device=dev)
        vp = torch.zeros_like(kp)
        outs["q"] = torch.zeros(M, nh * hs, device=dev)
        ops.linear(dtype=dt, M=M, N=nq, K=C, w0=wq, prologue=PRO_SCALED, epilogue=EPI_QKV_ROPE, x_packed=pk, x_ssq=ssq, eps=1e-5, row_pos=pos,
                   row_seq=seq, rope_cos=cos, rope_sin=sin, q_out=outs["q"], kv=ops.kv_geom(kp, vp, pt, nh, nkv, hs), workspace=ws, sum_order=order)
        outs["k"], outs["v"] = kp, vp
        torch.cuda.synchronize()
        return outs

    for epi in (EPI_RESIDUAL, EPI_STORE):
        g2env()
        y0, pk0, ssq0 = produce(0, epi)
        g2env(UA2_GEMM2_BMT=bmt)
        n0 = lib.ua2_debug_kernel_launches(b"gemm2")
        y1, pk1, ssq1 = produce(SUM_ORDER_FREE, epi)
        assert lib.ua2_debug_kernel_launches(b"gemm2") - n0 == 1
        _close(y1, y0, Kp, ("producer y", epi))
        assert not torch.equal(y1, y0)
        assert ((ssq1 - ssq0).abs() <= 1e-5 * ssq0.abs() + 1e-6).all(), float((ssq1 - ssq0).abs().max())
        _close(pk1.view(torch.bfloat16), pk0.view(torch.bfloat16), Kp, ("producer operand", epi))
        assert pk1.view(torch.bfloat16).float().abs().sum() > 0
    # consumers: the invariant kernels on the invariant hand-over vs the order-free kernel on the SAME hand-over
    g2env()
    want = consume(0, pk0, ssq0)
    g2env(UA2_GEMM2_BMT=bmt)
    n0 = lib.ua2_debug_kernel_launches(b"gemm2")
    got, again = consume(SUM_ORDER_FREE, pk0, ssq0), consume(SUM_ORDER_FREE, pk0, ssq0)
    assert lib.ua2_debug_kernel_launches(b"gemm2") - n0 == 6
    for k in want:
        assert torch.equal(got[k], again[k]), k
        _close(got[k], want[k], C, ("consumer", k))
        assert got[k].float().abs().sum() > 0
    assert not torch.equal(got["glu"], want["glu"])


@pytest.mark.parametrize("N,bmt", [(12296, 16), (12296, 8), (4096, 8)])
def test_argmax_partials_and_an_edge_column_count(N, bmt, g2env):
    """Round 6: UA2_EPI_STORE with per-16-column (max, index) partials on the order-free kernel — lm_head / audio_head at many rows
    (model_new.py:617,631-632 + :146-187 at topk = 1) — with a column count that is not a multiple of the wave's 64-column span
    (audio_head: 12 296 = 768 x 16 + 8) and a per-row forbidden prefix.  The partials must be EXACTLY the masked maxima of the logits
    the same launch stored (ties to the lowest index), and the logits agree with the invariant kernel's to summation noise."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_STORE, PRO_CAST, SUM_ORDER_FREE, lib
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(N + bmt)
    M, K = 700, 2048
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt, transposed=False)
    x = torch.randn(M, K, generator=g).to(dev)
    forbid = torch.randint(0, 4096, (M,), generator=g).to(torch.int32).to(dev)
    forbid[::3] = 0
    nb = (N + 15) // 16

    def run(order, **env):
        g2env(**env)
        y = torch.zeros(M, N, device=dev)
        pm, pi = torch.full((M, nb), float("nan"), device=dev), torch.full((M, nb), -7, dtype=torch.int32, device=dev)
        n0 = lib.ua2_debug_kernel_launches(b"gemm2")
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_STORE, x=x, y=y, part_max=pm, part_idx=pi, forbid=forbid,
                   workspace=ops.linear_workspace(dt, M, K, dev), sum_order=order)
        torch.cuda.synchronize()
        return y, pm, pi, lib.ua2_debug_kernel_launches(b"gemm2") - n0

    y0, pm0, pi0, n_inv = run(0)
    y1, pm1, pi1, n_free = run(SUM_ORDER_FREE, UA2_GEMM2_BMT=bmt)
    assert n_inv == 0 and n_free == 1
    _close(y1, y0, K, "logits")
    assert not torch.equal(y1, y0)
    # the partials of each launch describe that launch's own logits exactly
    for y, pm, pi in ((y0, pm0, pi0), (y1, pm1, pi1)):
        pad = torch.full((M, nb * 16 - N), float("-inf"), device=dev)
        cols = torch.arange(nb * 16, device=dev)[None]
        masked = torch.where(cols >= forbid[:, None].long(), torch.cat([y, pad], 1), torch.full_like(torch.cat([y, pad], 1), float("-inf")))
        t = masked.view(M, nb, 16)
        assert torch.equal(pm, t.max(-1).values)
        first = (t == t.max(-1, keepdim=True).values).float().argmax(-1) + 16 * torch.arange(nb, device=dev)[None]
        live = torch.isfinite(pm)
        assert torch.equal(pi[live].long(), first[live])
    # ... and the rows' winners agree wherever the invariant launch's margin exceeds the summation noise
    best0 = pm0.max(-1)
    top2 = torch.topk(pm0, 2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert clear.float().mean() > 0.9
    win0 = pi0.gather(1, best0.indices[:, None])[:, 0]
    win1 = pi1.gather(1, pm1.max(-1).indices[:, None])[:, 0]
    assert torch.equal(win0[clear], win1[clear])


def test_tail_split_of_a_grid_with_a_thin_last_round(g2env):
    """Round 6: 17 x 16 = 272 tiles of 256 x 256 on 256 CUs would run a second round of 16 workgroups; with scratch the launcher runs
    256 tiles whole and the last 16 as K slabs on a second launch + a combine over those tiles (the trunk's o- / down-projection at
    6272 rows: 300 tiles).  Deterministic, fp32-noise-close to the unsplit launch, identical outside the tail tiles, and the
    scaled-norm hand-over of the tail tiles comes from the combine."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, SUM_ORDER_FREE, lib
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(31)
    M, N, K = 4352, 4096, 4096
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt)
    x, res, nw = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev), (1.0 + 0.1 * torch.randn(N, generator=g)).to(dev)
    ws = ops.linear_workspace(dt, M, K, dev)

    def run(split, handover, **env):
        g2env(**env)
        y = torch.zeros(M, N, device=dev)
        sw = torch.full((8 * 16 * 256 * 256,), float("nan"), device=dev) if split else None
        kw = {}
        pk = ssq = None
        if handover:
            pk, ssq = ops.linear_workspace(dt, M, N, dev).zero_(), torch.zeros(M, N // 16, device=dev)
            kw = dict(y_norm_w=nw, y_packed=pk, y_ssq=ssq)
        n0 = lib.ua2_debug_kernel_launches(b"gemm2")
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res, workspace=ws, split_ws=sw,
                   sum_order=SUM_ORDER_FREE, **kw)
        torch.cuda.synchronize()
        return y, pk, ssq, lib.ua2_debug_kernel_launches(b"gemm2") - n0

    whole, _, _, n1 = run(False, False)
    split, _, _, n2 = run(True, False)
    again, _, _, _ = run(True, False)
    off, _, _, n3 = run(True, False, UA2_GEMM2_NO_TAIL=1)
    assert (n1, n2, n3) == (1, 2, 1)
    assert torch.equal(split, again) and torch.equal(off, whole) and not torch.isnan(split).any()
    diff = (split - whole).abs()
    assert 0 < diff.max().item() < 2e-5 * K ** 0.5
    changed = (diff > 0).view(M // 256, 256, N // 256, 256).any(3).any(1)        # which 256 x 256 tiles differ at all
    assert int(changed.sum()) <= 16, int(changed.sum())
    yh, pk, ssq, _ = run(True, True)
    yw, pkw, ssqw, _ = run(False, True)
    assert torch.equal(yh, split) and torch.equal(yw, whole)
    assert ((ssq - ssqw).abs() <= 1e-5 * ssqw.abs() + 1e-6).all()
    _close(pk.view(torch.bfloat16), pkw.view(torch.bfloat16), K, "tail hand-over operand")
    assert (ssq > 0).all()


@pytest.mark.parametrize("M,N,K,split", [(1000, 1536, 1536, True), (1000, 1536, 6144, True), (4100, 1536, 1536, False)])
def test_layernorm_handover_of_residual_launches(M, N, K, split, g2env):
    """Round 6 (ua2hip.h y_ln_w): a RESIDUAL launch under the order-free contract also writes the LayerNorm-ed, modulated, bf16-packed
    operand of the GEMM that follows — from the K-split combine where the launch runs as slabs (the DiT's o-projection and FF2 at one
    window: attention.py:345-349 -> :388-390, :401-405 -> the next block's :311-319), from a row pass otherwise.  y equals the plain
    launch's bit for bit; a consumer fed the packed operand (PRO_CAST) agrees with the same consumer running its own LayerNorm prep
    (PRO_NORM) on y to a bf16 rounding flip; ua2_linear_order_free_accepts agrees with what the launcher then does."""
    import ctypes
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, EPI_STORE, NORM_LAYERNORM, PRO_CAST, PRO_NORM, SUM_ORDER_FREE, lib
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(M + K)
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, generator=g)).to(dev)
    w = ops.pack_linear(mk(N, K, s=K ** -0.5), dt)
    wc = ops.pack_linear(mk(1024, N, s=N ** -0.5), dt)
    x, res, bias, gate = mk(M, K), mk(M, N, s=2.0), mk(N, s=0.3), mk(N, s=0.5)
    lw, lb = 1.0 + mk(N, s=0.2), mk(N, s=0.2)
    xp = ops.linear_workspace(dt, M, K, dev)
    g2env()
    ops.linear(dtype=dt, M=M, N=64, K=K, w0=ops.pack_linear(mk(64, K), dt), prologue=PRO_CAST, epilogue=EPI_STORE, x=x, y=torch.empty(M, 64, device=dev),
               workspace=xp)                                                  # leaves the packed operand in xp

    def run(ln):
        y = torch.zeros(M, N, device=dev)
        sw = torch.full((4 * M * N,), float("nan"), device=dev)
        pk = ops.linear_workspace(dt, M, N, dev).zero_() if ln else None
        kw = dict(y_ln=(lw, lb, 1e-6), y_packed=pk) if ln else {}
        a = ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x_packed=xp, y=y, resid=res, bias=bias, out_scale=gate,
                       split_ws=sw, sum_order=SUM_ORDER_FREE, launch=False, **kw)
        # without the hand-over 48 tiles of K = 1536 are too few for the kernel and too short for slabs: ua2_gemm.hip takes the launch
        takes = 0 if (not ln and M == 1000 and K == 1536) else 1
        assert lib.ua2_linear_order_free_accepts(ctypes.byref(a)) == takes
        n0 = lib.ua2_debug_kernel_launches(b"gemm2")
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x_packed=xp, y=y, resid=res, bias=bias, out_scale=gate,
                   split_ws=sw, sum_order=SUM_ORDER_FREE, **kw)
        torch.cuda.synchronize()
        assert lib.ua2_debug_kernel_launches(b"gemm2") - n0 == takes
        return y, pk, bool((~torch.isnan(sw)).any())

    y0, _, used0 = run(False)
    y1, pk, used1 = run(True)
    assert used1 == split                                                     # the K slabs ran (or, at 4100 rows, did not)
    if used0 == used1:
        assert torch.equal(y0, y1)                                            # same slabs, same combine arithmetic
    else:
        assert (y0 - y1).abs().max().item() < 2e-5 * K ** 0.5                 # the hand-over lowered the split's K threshold (K = 1536)
    z_pk, z_ln = torch.zeros(M, 1024, device=dev), torch.zeros(M, 1024, device=dev)
    ops.linear(dtype=dt, M=M, N=1024, K=N, w0=wc, prologue=PRO_CAST, epilogue=EPI_STORE, x_packed=pk, y=z_pk, sum_order=SUM_ORDER_FREE)
    ops.linear(dtype=dt, M=M, N=1024, K=N, w0=wc, prologue=PRO_NORM, epilogue=EPI_STORE, x=y1, norm_w=lw, norm_b=lb, eps=1e-6, norm_kind=NORM_LAYERNORM,
               y=z_ln, workspace=ops.linear_workspace(dt, M, N, dev), sum_order=SUM_ORDER_FREE)
    torch.cuda.synchronize()
    d = (z_pk - z_ln).abs()
    assert d.max().item() < 0.25 and d.double().pow(2).mean().sqrt().item() < 1e-2, (d.max().item(), d.double().pow(2).mean().sqrt().item())
    assert z_pk.abs().sum() > 0 and torch.isfinite(z_pk).all()
    # a launch the order-free kernel does not take refuses the hand-over instead of dropping it
    a = ops.linear(dtype=dt, M=64, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x[:64].contiguous(), y=y0[:64], resid=res[:64].contiguous(),
                   y_ln=(lw, lb, 1e-6), y_packed=pk, workspace=ops.linear_workspace(dt, 64, K, dev), sum_order=SUM_ORDER_FREE, launch=False)
    assert lib.ua2_linear_order_free_accepts(ctypes.byref(a)) == 0
    with pytest.raises(Exception, match="y_ln_w"):
        ops.linear(dtype=dt, M=64, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x[:64].contiguous(), y=y0[:64].contiguous(), resid=res[:64].contiguous(),
                   y_ln=(lw, lb, 1e-6), y_packed=pk, workspace=ops.linear_workspace(dt, 64, K, dev), sum_order=SUM_ORDER_FREE)
