"""Row invariance of the linear operator at the model's real shapes: a row's output must not depend on how many
other rows share the launch (rows per workgroup and launch geometry are functions of N, K and the
dtype only; per-row statistics are reduced in one fixed order).  Bit-exact comparisons of M = 1 launches against
the same rows inside M = 2, 5, 16, 17 and 40-row launches."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (N, K): backbone qkv-sized, down-proj, decoder qkv, head
    (5120, 3072), (3072, 8192), (3072, 2048), (12296 // 16 * 16, 2048),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,K", SHAPES)
def test_linear_rows_do_not_see_each_other(dtype, N, K):
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, PRO_NORM
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(N * 7 + K)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    w1 = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    p0, p1 = ops.pack_linear(w, dtype), ops.pack_linear(w1, dtype)
    nw = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dev)
    Mmax = 40
    x = torch.randn(Mmax, K, generator=g).to(dev)
    res = torch.randn(Mmax, N, generator=g).to(dev)

    def run(M, rows, pro, epi):
        xs = x[rows].contiguous()
        y = torch.empty(M, N, device=dev)
        kw = dict(dtype=dtype, M=M, N=N, K=K, w0=p0, prologue=pro, epilogue=epi, x=xs, y=y)
        if pro == PRO_NORM:
            kw.update(norm_w=nw, eps=1e-5)
        if epi == EPI_SWIGLU:
            kw.update(w1=p1)
        if epi == EPI_RESIDUAL:
            kw.update(resid=res[rows].contiguous())
        ops.linear(**kw)
        torch.cuda.synchronize()
        return y

    for pro, epi in [(PRO_CAST, EPI_STORE), (PRO_NORM, EPI_STORE), (PRO_NORM, EPI_SWIGLU), (PRO_CAST, EPI_RESIDUAL)]:
        single = torch.cat([run(1, slice(r, r + 1), pro, epi) for r in range(Mmax)])
        for M in (2, 5, 16, 17, 40):
            got = run(M, slice(0, M), pro, epi)
            assert torch.equal(got, single[:M]), (pro, epi, M, (got - single[:M]).abs().max().item())
        # a row keeps its value wherever it sits in the batch
        got = run(5, slice(7, 12), pro, epi)
        assert torch.equal(got, single[7:12])


@pytest.fixture
def linear_mode():
    """ua2_debug_force_general_linear: 2 = row-tiled decode kernel only, 4 / 5 = skinny / tiled large-M kernel."""
    from uniaudio2_amd._lib import lib

    def set_mode(m, bmt=None, no_glds=False):
        """bmt: row tiles per workgroup of the tiled kernel (UA2_GEMM_BMT: 8 = 128 x 128 tile, 4 = 64 x 128), None = its own choice;
        no_glds: the register-staged operand ring (UA2_GEMM_NO_GLDS) instead of the LDS-DMA ring."""
        lib.ua2_debug_force_general_linear(m)
        if bmt is None:
            os.environ.pop("UA2_GEMM_BMT", None)
        else:
            os.environ["UA2_GEMM_BMT"] = str(bmt)
        if no_glds:
            os.environ["UA2_GEMM_NO_GLDS"] = "1"
        else:
            os.environ.pop("UA2_GEMM_NO_GLDS", None)
    yield set_mode
    lib.ua2_debug_force_general_linear(0)
    os.environ.pop("UA2_GEMM_BMT", None)
    os.environ.pop("UA2_GEMM_NO_GLDS", None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,K", SHAPES + [(200, 72), (48, 4096)])
def test_large_m_kernel_is_bit_identical_to_the_decode_kernel(dtype, N, K, linear_mode):
    """ua2_gemm.hip reproduces ua2_gemv.hip's summation order: same bits for every row, every epilogue, at
    row counts from 1 to 300 (ragged last row-block, ragged last column tile, K not a multiple of the chunk)."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_GELU, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, PRO_NORM
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(N * 13 + K)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    w1 = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    p0, p1 = ops.pack_linear(w, dtype), ops.pack_linear(w1, dtype)
    nw, nb = (1.0 + 0.1 * torch.randn(K, generator=g)).to(dev), (0.1 * torch.randn(K, generator=g)).to(dev)
    Mmax = 300
    x = torch.randn(Mmax, K, generator=g).to(dev)
    res = torch.randn(Mmax, N, generator=g).to(dev)
    forbid = torch.randint(0, 8, (Mmax,), generator=g).int().to(dev)
    ws = ops.linear_workspace(dtype, Mmax, K, dev)
    npart = (N + 15) // 16

    def run(M, pro, epi, norm_kind=0):
        y = torch.zeros(M, N, device=dev)
        pm = torch.zeros(M, npart, device=dev)
        pi = torch.zeros(M, npart, dtype=torch.int32, device=dev)
        kw = dict(dtype=dtype, M=M, N=N, K=K, w0=p0, prologue=pro, epilogue=epi, x=x[:M], y=y, workspace=ws)
        if pro == PRO_NORM:
            kw.update(norm_w=nw, eps=1e-5, norm_kind=norm_kind, norm_b=nb)
        if epi == EPI_SWIGLU:
            kw.update(w1=p1)
        if epi == EPI_RESIDUAL:
            kw.update(resid=res[:M].contiguous())
        if epi == EPI_STORE:
            kw.update(part_max=pm, part_idx=pi, forbid=forbid)
        ops.linear(**kw)
        torch.cuda.synchronize()
        return y, pm, pi

    combos = [(PRO_CAST, EPI_STORE, 0), (PRO_NORM, EPI_STORE, 0), (PRO_NORM, EPI_SWIGLU, 0), (PRO_CAST, EPI_RESIDUAL, 0),
              (PRO_NORM, EPI_GELU, 2), (PRO_NORM, EPI_STORE, 1)]
    for pro, epi, nk in combos:
        for M in (1, 5, 40, 130, 300):
            linear_mode(2)
            ref = run(M, pro, epi, nk)
            # skinny form; 128-, 64- and 32-row tiled forms on the LDS-DMA ring; the register-staged ring (128- and 32-row)
            for mode, bmt, reg in ((4, None, False), (5, 8, False), (5, 4, False), (5, 2, False), (5, 8, True), (5, 2, True)):
                linear_mode(mode, bmt, reg)
                got = run(M, pro, epi, nk)
                for r, o, name in zip(ref, got, ("y", "part_max", "part_idx")):
                    assert torch.equal(r, o), (pro, epi, nk, M, mode, bmt, reg, name, (r.float() - o.float()).abs().max().item())
        linear_mode(0)                      # the launcher's own choice (cost model between the two forms)
        got = run(300, pro, epi, nk)
        for r, o in zip(ref, got):
            assert torch.equal(r, o)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("nh,nkv,hs,C", [(24, 8, 128, 3072), (32, 8, 64, 2048), (6, 6, 64, 384), (5, 5, 64, 320)])
def test_large_m_qkv_rope_cache_write_identical(dtype, nh, nkv, hs, C, linear_mode):
    """Fused RMSNorm + QKV + RoPE + paged KV-cache append through both kernels: q_out and the cache pages agree
    bit for bit (a 200-token prefill of one sequence)."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_QKV_ROPE, PRO_NORM
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(hs)
    M = 200
    nq = (nh + 2 * nkv) * hs
    mha_no_rope = nh == nkv                       # the codec's dense transformers: multi-head, no rotation, q|k|v bias (staged epilogue of its own)
    from uniaudio2_amd._lib import ROPE_NONE
    extra = dict(rope_mode=ROPE_NONE, bias=(0.1 * torch.randn(nq, generator=g)).to(dev)) if mha_no_rope else {}
    w = ops.pack_linear((torch.randn(nq, C, generator=g) * C ** -0.5).to(dev), dtype, **({} if mha_no_rope else dict(rope_head_size=hs)))
    x = torch.randn(M, C, generator=g).to(dev)
    nw = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    pos = torch.arange(M, dtype=torch.int32, device=dev)
    seq = torch.zeros(M, dtype=torch.int32, device=dev)
    ang = torch.rand(2048, hs // 2, generator=g) * 6.28
    cos, sin = ang.cos().to(dev), ang.sin().to(dev)
    pt = torch.arange(32, dtype=torch.int32, device=dev).flip(0).contiguous().view(1, 32)
    ws = ops.linear_workspace(dtype, M, C, dev)
    outs = []
    for mode, bmt in ((2, None), (4, None), (5, 8), (5, 4), (5, 2)):  # decode kernel, skinny, 128- / 64- / 32-row tiled (staged RoPE epilogue in all)
        linear_mode(mode, bmt)
        kp = torch.zeros(32, nkv, 64, hs, dtype=dtype, device=dev)
        vp = torch.zeros_like(kp)
        q = torch.zeros(M, nh * hs, device=dev)
        ops.linear(dtype=dtype, M=M, N=nq, K=C, w0=w, prologue=PRO_NORM, epilogue=EPI_QKV_ROPE, x=x, norm_w=nw,
                   row_pos=pos, row_seq=seq, rope_cos=cos, rope_sin=sin, q_out=q, kv=ops.kv_geom(kp, vp, pt, nh, nkv, hs),
                   workspace=ws, **extra)
        torch.cuda.synchronize()
        outs.append((q, kp, vp))
    for other in outs[1:]:
        for r, o in zip(outs[0], other):
            assert torch.equal(r, o)
    assert outs[0][1].float().abs().sum() > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("nh,nkv,hs", [(32, 8, 64), (4, 2, 32), (8, 4, 128)])
def test_local_attention_and_its_fusion_into_the_o_projection(dtype, nh, nkv, hs):
    """ua2_attn_local against a torch fp32 softmax(q k^T / sqrt(hs)) v on the cache contents, and the decode
    kernel's UA2_PRO_LOCAL_ATTN prologue (M = 1) against ua2_attn_local + UA2_PRO_CAST: same bits, so a sequence
    decodes identically alone (fused) and in a batch (stand-alone kernel)."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, PRO_LOCAL_ATTN
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(hs + nh)
    R, C = 5, nh * hs
    q = torch.randn(R, C, generator=g).to(dev)
    kp = torch.randn(R, nkv, 64, hs, generator=g).to(device=dev, dtype=dtype)
    vp = torch.randn(R, nkv, 64, hs, generator=g).to(device=dev, dtype=dtype)
    pt = torch.arange(R, dtype=torch.int32, device=dev).flip(0).contiguous().view(R, 1)
    pos = torch.tensor([0, 3, 7, 1, 5], dtype=torch.int32, device=dev)
    geom = ops.kv_geom(kp, vp, pt, nh, nkv, hs)
    y = torch.zeros(R, C, device=dev)
    ops.attn_local(dtype=dtype, R=R, q=q, row_pos=pos, row_seq=None, kv=geom, y=y)
    torch.cuda.synchronize()
    G = nh // nkv
    for r in range(R):
        n = int(pos[r]) + 1
        page = int(pt[r, 0])
        k = kp[page, :, :n].float().repeat_interleave(G, dim=0)          # (nh, n, hs)
        v = vp[page, :, :n].float().repeat_interleave(G, dim=0)
        qq = q[r].view(nh, 1, hs)
        p = torch.softmax((qq @ k.transpose(1, 2)) / hs ** 0.5, dim=-1)
        ref = (p @ v).reshape(C)
        assert (y[r] - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item()), r
    # fusion: row by row through the O-projection
    w = ops.pack_linear((torch.randn(C, C, generator=g) * C ** -0.5).to(dev), dtype)
    res = torch.randn(R, C, generator=g).to(dev)
    for r in range(R):
        a = torch.zeros(1, C, device=dev)
        b = torch.zeros(1, C, device=dev)
        ops.linear(dtype=dtype, M=1, N=C, K=C, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=y[r:r + 1].contiguous(), y=a,
                   resid=res[r:r + 1].contiguous())
        ops.linear(dtype=dtype, M=1, N=C, K=C, w0=w, prologue=PRO_LOCAL_ATTN, epilogue=EPI_RESIDUAL, x=q[r:r + 1].contiguous(), y=b,
                   resid=res[r:r + 1].contiguous(), row_pos=pos[r:r + 1].contiguous(),
                   row_seq=torch.tensor([r], dtype=torch.int32, device=dev), kv=geom)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (r, (a - b).abs().max().item())


def _unpack_operand(pk, M, K):
    """[ceil(M/16)][K/32][64 lanes][8] bf16 fragment order -> [M, K] (element (m, k): chunk k // 32, lane (k % 32) // 8 * 16 + m % 16)."""
    t = pk.view(-1, K // 32, 4, 16, 8)                   # [mt][chunk][g][m & 15][e]
    return t.permute(0, 3, 1, 2, 4).reshape(-1, K)[:M]


def test_scaled_norm_handover_is_kernel_and_row_count_invariant(linear_mode):
    """Round 3's RMSNorm hand-over (include/ua2hip.h UA2_PRO_SCALED / y_norm_w): the producer's three outputs (fp32 y,
    RNE_bf16(y * w_next) in row-major or fragment order, per-16-column sums of squares) and the consumer's result
    rstd * (operand W^T) are bit-identical whichever kernel the row count selects (decode kernel with the row-major
    hand-over; skinny / tiled kernels with the packed one) and whatever rows share the launch; and they are the function the
    contract states, against a plain torch evaluation."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, lib
    PRO_SCALED = 4
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(31)
    C, N2, I = 3072, 2048, 8192
    Mmax = 70
    w_o = ops.pack_linear((torch.randn(C, C, generator=g) * C ** -0.5).to(dev), dt)          # the producer: a residual projection
    W2 = (torch.randn(N2, C, generator=g) * C ** -0.5).to(dev)
    w_c = ops.pack_linear(W2, dt)                                                              # consumers: STORE and SWIGLU
    Wg, Wu = (torch.randn(I, C, generator=g) * C ** -0.5).to(dev), (torch.randn(I, C, generator=g) * C ** -0.5).to(dev)
    w_g, w_u = ops.pack_linear(Wg, dt), ops.pack_linear(Wu, dt)
    nw = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev)
    x = torch.randn(Mmax, C, generator=g).to(dev)
    res = (3.0 * torch.randn(Mmax, C, generator=g)).to(dev)

    def produce(rows, mode, packed):
        M = rows.stop - rows.start
        linear_mode(mode)
        y = torch.empty(M, C, device=dev)
        ssq = torch.zeros(M, C // 16, device=dev)
        kw = dict(dtype=dt, M=M, N=C, K=C, w0=w_o, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x[rows].contiguous(), y=y,
                  resid=res[rows].contiguous(), y_norm_w=nw, y_ssq=ssq, workspace=ops.linear_workspace(dt, M, C, dev))
        if packed:
            pk = torch.zeros((M + 15) // 16 * 16 * C, dtype=dt, device=dev)
            ops.linear(**kw, y_packed=pk)
            h = _unpack_operand(pk, M, C)
        else:
            h = torch.zeros(M, C, dtype=dt, device=dev)
            ops.linear(**kw, y_h=h)
            pk = None
        torch.cuda.synchronize()
        return y, ssq, h, pk

    def consume(M, mode, h, pk, ssq, swiglu):
        linear_mode(mode)
        N = I if swiglu else N2
        out = torch.empty(M, N, device=dev)
        kw = dict(dtype=dt, M=M, N=N, K=C, w0=w_g if swiglu else w_c, w1=w_u if swiglu else None, prologue=PRO_SCALED,
                  epilogue=EPI_SWIGLU if swiglu else EPI_STORE, y=out, x_ssq=ssq, eps=1e-5)
        if pk is not None:
            ops.linear(**kw, x_packed=pk)
        else:
            ops.linear(**kw, x_h=h.contiguous())
        torch.cuda.synchronize()
        return out

    # single-row reference runs through the decode kernel and the row-major hand-over
    singles = [produce(slice(r, r + 1), 2, False) for r in range(Mmax)]
    y1 = torch.cat([s[0] for s in singles]); q1 = torch.cat([s[1] for s in singles]); h1 = torch.cat([s[2] for s in singles])
    c1 = torch.cat([consume(1, 2, singles[r][2], None, singles[r][1], False) for r in range(Mmax)])
    g1 = torch.cat([consume(1, 2, singles[r][2], None, singles[r][1], True) for r in range(0, Mmax, 9)])
    # the contract, in plain torch (fp32 accumulate; bf16 inputs): y = resid + bf16(x) Wo^T is checked by the other tests; here
    # the hand-over of THAT y and the scaled consumer
    assert torch.equal(h1, (y1 * nw).to(dt))
    torch.testing.assert_close(q1, (y1.double() ** 2).view(Mmax, C // 16, 16).sum(-1).float(), rtol=2e-6, atol=0)
    rstd = torch.rsqrt((y1.double() ** 2).mean(-1, keepdim=True) + 1e-5)
    want = (rstd * (h1.double() @ W2.to(dt).double().t())).float()
    torch.testing.assert_close(c1, want, rtol=0, atol=2e-3)
    hg = h1[::9].double()
    want_g = (torch.nn.functional.silu(rstd[::9] * (hg @ Wg.to(dt).double().t())) * (rstd[::9] * (hg @ Wu.to(dt).double().t()))).float()
    torch.testing.assert_close(g1, want_g, rtol=0, atol=4e-3)
    for M in (2, 16, 17, 40, 70):
        for mode, packed in ((2, False), (4, True), (5, True)) if M <= 16 else ((4, True), (5, True)):
            y, ssq, h, pk = produce(slice(0, M), mode, packed)
            assert torch.equal(y, y1[:M]) and torch.equal(ssq, q1[:M]) and torch.equal(h, h1[:M]), (M, mode)
            c = consume(M, mode, h, pk, ssq, False)
            assert torch.equal(c, c1[:M]), (M, mode, (c - c1[:M]).abs().max().item())
    y, ssq, h, pk = produce(slice(0, 64), 4, True)
    gm = consume(64, 4, h, pk, ssq, True)
    assert torch.equal(gm[::9], g1[:8])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,K", [(1536, 1536), (256, 2048), (128, 3072), (64, 72), (192, 1056), (64, 4096)])
def test_row_tile_prep_equals_the_per_row_prep(dtype, N, K, linear_mode):
    """The operand prep of the many-row kernels, one 16-row tile per workgroup (round 4), against the per-row kernel
    (UA2_GEMM_OLD_PREP): same statistics order, same rounding, so the GEMM behind it returns the same bits — cast, RMSNorm
    (both flavours) and LayerNorm prologues, ragged row counts, K with a partial last chunk."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_STORE, PRO_CAST, PRO_NORM
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(K + N)
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dtype)
    nw, nb = (1.0 + 0.2 * torch.randn(K, generator=g)).to(dev), (0.2 * torch.randn(K, generator=g)).to(dev)
    x = (torch.randn(1000, K, generator=g) * 3.0 + 0.5).to(dev)
    try:
        for M in (1000, 17, 77):
            for pro, nk in ((PRO_CAST, 0), (PRO_NORM, 0), (PRO_NORM, 1), (PRO_NORM, 2)):
                outs = []
                # the per-row kernel, the 16-rows-per-workgroup form (taken from 2048 rows up) and the 4-rows form (below), each forced
                for old, min_rows in ((True, "1"), (False, "1"), (False, "1000000")):
                    if old:
                        os.environ["UA2_GEMM_OLD_PREP"] = "1"
                    else:
                        os.environ.pop("UA2_GEMM_OLD_PREP", None)
                    os.environ["UA2_GEMM_PREP16_MIN_ROWS"] = min_rows
                    os.environ["UA2_GEMM_PREP4_MIN_ROWS"] = "1"       # (the launcher keeps the per-row kernel below 960 rows)
                    linear_mode(5)
                    y = torch.zeros(M, N, device=dev)
                    ops.linear(dtype=dtype, M=M, N=N, K=K, w0=w, prologue=pro, epilogue=EPI_STORE, x=x[:M].contiguous(), y=y,
                               norm_w=nw, norm_b=nb, norm_kind=nk, eps=1e-5, workspace=ops.linear_workspace(dtype, M, K, dev))
                    torch.cuda.synchronize()
                    outs.append(y)
                assert torch.equal(outs[0], outs[1]), (M, pro, nk, "16-row form", (outs[0] - outs[1]).abs().max().item())
                assert torch.equal(outs[0], outs[2]), (M, pro, nk, "4-row form", (outs[0] - outs[2]).abs().max().item())
                assert outs[1].abs().sum().item() > 0
    finally:
        os.environ.pop("UA2_GEMM_OLD_PREP", None)
        os.environ.pop("UA2_GEMM_PREP16_MIN_ROWS", None)
        os.environ.pop("UA2_GEMM_PREP4_MIN_ROWS", None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_staged_epilogue_equals_the_per_element_epilogue(dtype, linear_mode):
    """Round 4: the tiled kernel's epilogues (STORE / RESIDUAL / SWIGLU / GELU) walk a wave's patch through LDS in 16-byte
    pieces.  Same operations on every value as the per-element form (UA2_GEMM_OLD_EPI), so the same bits: every output of every
    option the epilogues have (bias, LayerScale, both activations of each kind, packed operand hand-off, the scaled-norm
    hand-over on both sides), on every tile size and both operand rings, at ragged row counts."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import (EPI_GELU, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, GATE_SIGMOID_SECOND, GELU_TANH, PRO_CAST, PRO_NORM)
    PRO_SCALED = 4
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(77)
    N, K = 1536, 1056
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, generator=g)).to(dev)
    p0, p1 = ops.pack_linear(mk(N, K, s=K ** -0.5), dtype), ops.pack_linear(mk(N, K, s=K ** -0.5), dtype)
    bias, bias1, gate, nw_next, nw = mk(N, s=0.3), mk(N, s=0.3), mk(N, s=0.5), 1.0 + mk(N, s=0.2), 1.0 + mk(K, s=0.1)
    Mmax = 1000
    x, res = mk(Mmax, K), mk(Mmax, N, s=2.0)

    def run(M, epi, old, **opt):
        if old:
            os.environ["UA2_GEMM_OLD_EPI"] = "1"
        else:
            os.environ.pop("UA2_GEMM_OLD_EPI", None)
        outs = {}
        kw = dict(dtype=dtype, M=M, N=N, K=K, w0=p0, epilogue=epi, workspace=ops.linear_workspace(dtype, M, K, dev))
        if opt.get("scaled"):
            xp, xs = scaled_operand(M)
            kw.update(prologue=PRO_SCALED, x_ssq=xs, x_packed=xp, eps=1e-5)
            kw.pop("workspace")
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
            outs["packed"] = torch.zeros((M + 15) // 16 * 16 * N, dtype=dtype, device=dev)
            kw.update(y_packed=outs["packed"])
        if opt.get("handover"):
            outs["ssq"] = torch.zeros(M, N // 16, device=dev)
            outs["h"] = torch.zeros(M, N, dtype=dtype, device=dev)
            kw.update(y_norm_w=nw_next, y_ssq=outs["ssq"], y_h=outs["h"], ldh=N)
        ops.linear(**kw)
        torch.cuda.synchronize()
        return outs

    pk = ops.pack_linear(mk(K, K, s=K ** -0.5), dtype)
    made = {}

    def scaled_operand(M):
        """operand of a UA2_PRO_SCALED launch as a producer hands it over: fragment order + per-16-column sums of squares"""
        if M not in made:
            os.environ["UA2_GEMM_OLD_EPI"] = "1"
            linear_mode(5, 4, False)
            xp = torch.zeros((M + 15) // 16 * 16 * K, dtype=dtype, device=dev)
            xs = torch.zeros(M, K // 16, device=dev)
            ops.linear(dtype=dtype, M=M, N=K, K=K, w0=pk, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x[:M].contiguous(),
                       y=torch.empty(M, K, device=dev), resid=x[:M].contiguous(), y_norm_w=nw, y_ssq=xs, y_packed=xp,
                       workspace=ops.linear_workspace(dtype, M, K, dev))
            torch.cuda.synchronize()
            made[M] = (xp, xs)
        return made[M]

    cases = [(EPI_STORE, dict()), (EPI_STORE, dict(bias=True)),
             (EPI_RESIDUAL, dict()), (EPI_RESIDUAL, dict(bias=True, gate=True)),
             (EPI_SWIGLU, dict()), (EPI_SWIGLU, dict(bias=True, act=GATE_SIGMOID_SECOND, packed=True)), (EPI_SWIGLU, dict(packed=True, y=False)),
             (EPI_GELU, dict(bias=True)), (EPI_GELU, dict(act=GELU_TANH, packed=True, y=False))]
    if dtype == torch.bfloat16:
        cases += [(EPI_STORE, dict(handover=True)), (EPI_RESIDUAL, dict(handover=True, gate=True, packed=True)),
                  (EPI_STORE, dict(scaled=True)), (EPI_SWIGLU, dict(scaled=True))]
    try:
        for epi, opt in cases:
            for M in (1000, 200, 77):
                for bmt, reg in ((8, False), (4, False), (2, False), (8, True), (2, True)):
                    if opt.get("scaled"):
                        scaled_operand(M)
                    linear_mode(5, bmt, reg)
                    want, got = run(M, epi, True, **opt), run(M, epi, False, **opt)
                    assert want.keys() == got.keys() and len(want) > 0
                    for k in want:
                        assert torch.equal(want[k].view(torch.int16 if want[k].dtype == torch.bfloat16 else torch.int32),
                                           got[k].view(torch.int16 if got[k].dtype == torch.bfloat16 else torch.int32)), (epi, opt, M, bmt, reg, k)
                    assert any(t.float().abs().sum().item() > 0 for t in got.values())
    finally:
        os.environ.pop("UA2_GEMM_OLD_EPI", None)


def test_k_split_of_long_k_residual_launches(linear_mode):
    """ua2hip.h split_ws (ABI v7): a RESIDUAL launch with a long K and a small grid may run its K range as up to four slabs side
    by side and combine them in index order.  Opt-in (scratch given), deterministic, equal to the unsplit launch up to the
    rounding of a different summation order; without scratch, with too little scratch, with the hook UA2_GEMM_NO_KSPLIT or on
    shapes outside the rule nothing changes."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST
    dev, dt = torch.device("cuda"), torch.bfloat16
    g = torch.Generator().manual_seed(9)
    M, N, K = 1000, 1536, 6144
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt)
    x, res = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
    bias, gate = (0.3 * torch.randn(N, generator=g)).to(dev), (0.5 * torch.randn(N, generator=g)).to(dev)

    def run(split_elems, **env):
        for k, v in env.items():
            os.environ[k] = v
        try:
            linear_mode(5)
            y = torch.zeros(M, N, device=dev)
            sw = torch.full((split_elems,), float("nan"), device=dev) if split_elems else None
            ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res, bias=bias, out_scale=gate,
                       workspace=ops.linear_workspace(dt, M, K, dev), split_ws=sw)
            torch.cuda.synchronize()
            return y
        finally:
            for k in env:
                os.environ.pop(k, None)

    base = run(0)
    split4, again = run(4 * M * N), run(4 * M * N)
    assert torch.equal(split4, again)                                   # deterministic
    assert not torch.isnan(split4).any()
    err = (split4 - base).abs().max().item()
    assert 0 < err < 2e-5 * K ** 0.5, err                              # another summation order, nothing more
    split2 = run(2 * M * N + 7)                                         # scratch for two slabs only
    assert torch.equal(split2, run(2 * M * N + 7)) and (split2 - base).abs().max().item() < 2e-5 * K ** 0.5
    assert torch.equal(run(M * N), base)                                # not even two slabs fit: the unsplit launch
    assert torch.equal(run(4 * M * N, UA2_GEMM_NO_KSPLIT="1"), base)
    # outside the rule (short K): scratch or not, the same bits
    K2 = 1536
    w2 = ops.pack_linear((torch.randn(N, K2, generator=g) * K2 ** -0.5).to(dev), dt)
    outs = []
    for sw in (None, torch.empty(4 * M * N, device=dev)):
        linear_mode(5)
        y = torch.zeros(M, N, device=dev)
        ops.linear(dtype=dt, M=M, N=N, K=K2, w0=w2, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x[:, :K2].contiguous(), y=y, resid=res,
                   workspace=ops.linear_workspace(dt, M, K2, dev), split_ws=sw)
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,bmt", [(6272, 1024, 3072, 8), (2048, 3072, 8192, 4), (1000, 1536, 6144, 2), (333, 384, 1056, 2)])
def test_tiled_gemm_ring_is_repeatable(M, N, K, bmt, linear_mode):
    """Race screen for the LDS-DMA operand ring (hand-counted vmcnt + raw s_barrier): 25 launches of the same problem on a
    busy device give the same bits every time, and those bits are the register-staged ring's (a different synchronisation
    structure).  An early fragment read or a late ring refill would show up as rare wrong tiles."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST
    dev = torch.device("cuda")
    dt = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt)
    x = torch.randn(M, K, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    ws = ops.linear_workspace(dt, M, K, dev)
    noise = torch.randn(4096, 4096, device=dev)

    def run(reg):
        linear_mode(5, bmt, reg)
        y = torch.empty(M, N, device=dev)
        ops.linear(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y, resid=res, workspace=ws)
        return y

    ref = run(True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for i in range(25):
        with torch.cuda.stream(side):              # unrelated traffic on another stream: shifts arrival times in the ring
            noise = noise * 1.0001
        got = run(False)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), (i, (got - ref).abs().max().item())


@pytest.mark.parametrize("N", [3072, 2048])
def test_range_split_of_the_k8192_down_projection_keeps_every_bit(N, linear_mode):
    """Round 6 (ua2hip.h range_ws, csrc/ua2_skinny.hip rsplit_*): at 33-64 rows the K = 8192 RESIDUAL launches of the row-invariant plan
    may run each of the decode kernel's 16 K ranges on its own workgroups and add the range partials in range order in a second launch.
    y, the scaled-norm hand-over (fragment-order operand, per-16-column sums of squares) and a consumer fed with them must equal, bit for
    bit, the launch without the scratch AND the rows' single-row runs through the decode kernel; the counter says the split really ran."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_RESIDUAL, PRO_CAST, lib
    dev, dt, K = torch.device("cuda"), torch.bfloat16, 8192
    g = torch.Generator().manual_seed(N)
    w = ops.pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).to(dev), dt)
    nw = (1.0 + 0.2 * torch.randn(N, generator=g)).to(dev)
    x = torch.randn(64, K, generator=g).to(dev)
    res = (2.0 * torch.randn(64, N, generator=g)).to(dev)
    rws = torch.empty(16 * 64 * N, device=dev)

    def run(rows, scratch, mode=0):
        M = rows.stop - rows.start
        linear_mode(mode)
        y = torch.empty(M, N, device=dev)
        ssq = torch.zeros(M, N // 16, device=dev)
        pk = torch.zeros((M + 15) // 16 * 16 * N, dtype=dt, device=dev)
        h = torch.zeros(M, N, dtype=dt, device=dev)
        kw = dict(dtype=dt, M=M, N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x[rows].contiguous(), y=y,
                  resid=res[rows].contiguous(), y_norm_w=nw, y_ssq=ssq, workspace=ops.linear_workspace(dt, M, K, dev))
        if M > 16:
            kw["y_packed"] = pk
        else:
            kw["y_h"] = h
        n0 = lib.ua2_debug_kernel_launches(b"rsplit")
        ops.linear(**kw, range_ws=rws if scratch else None)
        torch.cuda.synchronize()
        return y, ssq, (_unpack_operand(pk, M, N) if M > 16 else h), lib.ua2_debug_kernel_launches(b"rsplit") - n0

    singles = [run(slice(r, r + 1), False, 2) for r in range(64)]
    y1 = torch.cat([s[0] for s in singles]); q1 = torch.cat([s[1] for s in singles]); h1 = torch.cat([s[2] for s in singles])
    for M in (33, 48, 64):
        y0, q0, h0, n_plain = run(slice(0, M), False)
        ys, qs, hs, n_split = run(slice(0, M), True)
        assert n_plain == 0 and n_split == 1, (M, n_plain, n_split)
        for got in ((y0, q0, h0), (ys, qs, hs)):
            assert torch.equal(got[0], y1[:M]) and torch.equal(got[1], q1[:M]) and torch.equal(got[2], h1[:M]), M
    for M in (17, 32):                                       # outside the form (one or two row tiles): the scratch is ignored
        y, q, h, n = run(slice(0, M), True)
        assert n == 0 and torch.equal(y, y1[:M]) and torch.equal(q, q1[:M]) and torch.equal(h, h1[:M]), M
