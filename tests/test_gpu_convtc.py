"""Decode-side convolutions on time-major split planes (ua2_conv1d_tc, round 4) against the round-2 bf16 x 3 kernel they
replace on the decode path, and the pipelined LDS-DMA form against the plain form.

Bars.  (1) On inputs that are exactly representable as hi + lo (16 significant bits — what every decode-side layer receives),
ua2_conv1d_tc computes the same MFMA sequence per accumulator as ua2_conv1d precision 1 and the same epilogue operations, and
then rounds its output to hi / lo planes: so its planes must EQUAL ops.tc_pack(old kernel's fp32 output) bit for bit
(torch.equal) — conv, transposed-conv phases, repeat-upsampling, residual, fused residual unit.  The old kernel is pinned on
torch / the reference's goldens in tests/test_gpu_conv.py and tests/test_gpu_codec.py.  (2) Pipelined == plain, torch.equal.
The end-to-end 1e-4 RMS bar of the decoded waveform lives in tests/test_gpu_codec.py."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _representable(x):
    """Round an fp32 tensor to the 16 significant bits the planes carry (hi + lo)."""
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi + lo


def test_tc_pack_unpack_round_trip():
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(2, 96, 333, generator=g) * torch.logspace(-3, 3, 96).view(1, 96, 1)).cuda()
    t = ops.tc_pack(x)
    assert t.planes.shape == (2, 2, 333, 96) and t.shape == (2, 96, 333)
    y = ops.tc_unpack(t)
    assert torch.equal(y, _representable(x))
    assert torch.equal(t.hi.transpose(1, 2).float(), x.to(torch.bfloat16).float())
    assert ((y - x).abs() <= x.abs() * 2.0 ** -17 + 1e-30).all()          # 16+ significant bits
    assert torch.equal(ops.tc_unpack(ops.tc_pack(y)), y)                   # idempotent


def _run_pair(Cin, Cout, K, dil, T, B, residual, fused, rep, phases, trim, per_channel_alpha, seed, variant):
    """Same problem through the old kernel (fp32 [C][T]) and ua2_conv1d_tc; returns (packed old output, tc output)."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import ACT_PRELU
    g = torch.Generator().manual_seed(seed)
    x = _representable(torch.randn(B, Cin, T, generator=g)).cuda()
    rows = Cout * phases
    w = (torch.randn(rows, Cin, K, generator=g) / (Cin * K) ** 0.5).cuda()
    hi, lo = ops.pack_conv_weight_x3(w)
    bias = torch.randn(Cout, generator=g).cuda()
    alpha = (torch.rand(Cout, generator=g) * 0.5 if per_channel_alpha else torch.tensor([0.2])).cuda()
    Tout = T * rep * phases - 2 * trim if phases > 1 else T * rep
    kw = dict(dilation=dil, pad_left=dil * (K - 1), Tout=Tout, bias=bias, post_act=ACT_PRELU, post_alpha=alpha, in_repeat=rep,
              out_phases=phases, out_trim_left=trim)
    okw, tkw = dict(kw), dict(kw)
    if fused:
        w2 = (torch.randn(Cout, Cout, 1, generator=g) / Cout ** 0.5).cuda()
        b2, a2 = torch.randn(Cout, generator=g).cuda(), torch.tensor([0.3]).cuda()
        okw.update(residual=x, fused2=(*ops.pack_conv_weight_x3(w2), b2, a2))
        tkw.update(fused2=(*ops.pack_conv_weight_x3(ops.tc_w2_order(w2)), b2, a2))    # the tc kernels' K order of the 1 x 1 conv
    elif residual:
        r = _representable(torch.randn(B, Cout, Tout, generator=g)).cuda()
        okw.update(residual=r)
        tkw.update(residual=ops.tc_pack(r))
    old = ops.conv1d(x, hi, K, Cout, w_lo=lo, **okw)
    new = ops.conv1d_tc(ops.tc_pack(x), hi, lo, K, Cout, variant=variant, **tkw)
    torch.cuda.synchronize()
    assert torch.isfinite(old).all()
    return ops.tc_pack(old), new


PLAIN_CASES = [
    # Cin, Cout, K, dil, T, B, residual, fused, rep, phases, trim, per-channel alpha
    (32, 32, 7, 9, 1000, 2, False, True, 1, 1, 0, False), (64, 64, 7, 1, 333, 2, False, True, 1, 1, 0, False),
    (128, 128, 7, 5, 257, 1, False, True, 1, 1, 0, False), (512, 512, 7, 3, 150, 1, False, False, 1, 1, 0, False),
    (256, 256, 1, 1, 777, 2, True, False, 1, 1, 0, True), (64, 32, 7, 1, 2049, 1, False, False, 2, 1, 0, False),
    (128, 64, 2, 1, 513, 2, False, False, 1, 4, 0, False), (1024, 512, 2, 1, 50, 1, False, False, 1, 3, 0, False),
    (64, 32, 2, 1, 700, 1, False, False, 1, 2, 1, False), (96, 160, 3, 2, 70, 3, True, False, 1, 1, 0, True),
    (32, 64, 5, 1, 37, 1, False, False, 1, 1, 0, False), (160, 96, 1, 1, 64, 2, False, False, 1, 1, 0, False),
]


def _same(got, want, fused):
    """Bit-equal planes; a fused unit's 1 x 1 conv reduces its 32-channel chunks in another order than the round-2 kernel
    (tc_w2_order), so there the values agree to fp32 summation noise (a few units of the 16-bit output grid at worst)."""
    if not fused:
        assert torch.equal(got.planes, want.planes), f"max diff {(got.planes.float() - want.planes.float()).abs().max().item():.3e}"
        return
    from uniaudio2_amd import ops
    a, b = ops.tc_unpack(got), ops.tc_unpack(want)
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() <= 2.0 ** -14 * scale, f"max diff {(a - b).abs().max().item():.3e} vs scale {scale:.3e}"
    assert (got.planes[0] != want.planes[0]).float().mean().item() < 0.01      # the hi plane agrees almost everywhere


@pytest.mark.parametrize("case", PLAIN_CASES)
def test_plain_tc_kernel_equals_packed_output_of_the_round2_kernel(case):
    want, got = _run_pair(*case, seed=sum(case[:5]), variant=1)
    _same(got, want, fused=case[7])


def test_tc_waveform_output_is_fp32_channel_major():
    """The last decoder layer (32 -> 1 channels, k = 7) writes the fp32 waveform directly: equal to the round-2 kernel's."""
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(9)
    x = _representable(torch.randn(2, 32, 5000, generator=g)).cuda()
    w = (torch.randn(1, 32, 7, generator=g) / 15).cuda()
    hi, lo = ops.pack_conv_weight_x3(w)
    b = torch.randn(1, generator=g).cuda()
    old = ops.conv1d(x, hi, 7, 1, pad_left=6, Tout=5000, bias=b, w_lo=lo)
    for variant in (1, 2):
        new = ops.conv1d_tc(ops.tc_pack(x), hi, lo, 7, 1, pad_left=6, Tout=5000, bias=b, out_f32=True, variant=variant)
        assert new.shape == (2, 1, 5000) and torch.equal(new, old), variant


PIPE_CASES = [
    # every instantiation of the pipelined kernel: fused units (32 / 64 / 128 channels, all dilations of the decoder), wide k7
    # convs, the PostProcessor conv (repeat-upsampled input), up-sampler phases with 4 and 2 channel groups per unit, 1 x 1 convs
    # with and without residual planes; lengths that end inside a tile, one and many tiles per workgroup, batch > 1
    (32, 32, 7, 9, 1000, 2, False, True, 1, 1, 0, False), (32, 32, 7, 1, 240000, 1, False, True, 1, 1, 0, False),
    (64, 64, 7, 1, 4097, 2, False, True, 1, 1, 0, False), (64, 64, 7, 9, 120000, 1, False, True, 1, 1, 0, False),
    (64, 64, 7, 5, 63, 1, False, True, 1, 1, 0, False), (128, 128, 7, 5, 700, 2, False, True, 1, 1, 0, False),
    (128, 128, 7, 9, 30000, 1, False, True, 1, 1, 0, False), (128, 128, 7, 3, 7, 1, False, True, 1, 1, 0, False),
    (512, 512, 7, 3, 1500, 1, False, False, 1, 1, 0, False), (256, 256, 7, 9, 7500, 1, False, False, 1, 1, 0, True),
    (256, 512, 7, 7, 333, 2, False, False, 1, 1, 0, False), (32, 32, 7, 1, 2049, 1, False, False, 2, 1, 0, False),
    (1024, 512, 2, 1, 500, 1, False, False, 1, 3, 0, False), (512, 256, 2, 1, 1500, 1, False, False, 1, 5, 0, False),
    (128, 64, 2, 1, 30000, 1, False, False, 1, 4, 0, False), (64, 32, 2, 1, 4100, 2, False, False, 1, 2, 0, False),
    (64, 32, 2, 1, 700, 1, False, False, 1, 2, 1, False), (512, 512, 1, 1, 1500, 1, True, False, 1, 1, 0, False),
    (256, 256, 1, 1, 7500, 1, True, False, 1, 1, 0, True), (256, 128, 1, 1, 777, 2, False, False, 1, 1, 0, False),
]


@pytest.mark.parametrize("case", PIPE_CASES)
def test_pipelined_tc_kernel_is_bit_identical_to_plain(case):
    """LDS-DMA windows, fragment read-ahead, weights refilled in place, residual from the LDS window — same split, same
    products, same summation order, same epilogue operations as the plain kernel: torch.equal on both planes."""
    from uniaudio2_amd import ops
    ref_old, plain = _run_pair(*case, seed=sum(case[:5]) + 1, variant=1)
    _, pipe = _run_pair(*case, seed=sum(case[:5]) + 1, variant=2)
    assert torch.equal(pipe.planes, plain.planes), f"max diff {(pipe.planes.float() - plain.planes.float()).abs().max().item():.3e}"
    _same(plain, ref_old, fused=case[7])
    # repeatable (the hand-counted LDS-DMA waits are a race screen's business: same bits on every launch)
    for _ in range(3):
        _, again = _run_pair(*case, seed=sum(case[:5]) + 1, variant=2)
        assert torch.equal(again.planes, pipe.planes)


BIG_CASES = [
    # the "one big tile per workgroup" kernel: fused 32- / 64-channel units and the 32 -> 32 conv behind the repeat-upsampling;
    # every dilation of the decoder, lengths that are / are not whole tiles, batch 2
    (32, 32, 7, 9, 240000, 1, False, True, 1, 1, 0, False), (32, 32, 7, 1, 100001, 1, False, True, 1, 1, 0, True),
    (32, 32, 7, 5, 50000, 2, False, True, 1, 1, 0, False), (64, 64, 7, 9, 120000, 1, False, True, 1, 1, 0, False),
    (64, 64, 7, 3, 49153, 1, False, True, 1, 1, 0, True), (64, 64, 7, 7, 25000, 2, False, True, 1, 1, 0, False),
    (64, 64, 7, 1, 60000, 1, False, True, 1, 1, 0, False), (32, 32, 7, 1, 120000, 1, False, False, 2, 1, 0, False),
    (32, 32, 7, 1, 49000, 2, False, False, 2, 1, 0, True),
]


@pytest.mark.parametrize("case", BIG_CASES)
def test_big_tile_tc_kernel_is_bit_identical_to_plain(case):
    """convtc_big_kernel (a wave owns all output rows; filter through a register ring; all windows of a 512 / 1024-step tile
    resident; the fused 1 x 1 conv fed from the accumulator registers) against the plain kernel: torch.equal, and repeatable."""
    ref_old, plain = _run_pair(*case, seed=sum(case[:5]) + 2, variant=1)
    _, big = _run_pair(*case, seed=sum(case[:5]) + 2, variant=3)
    assert torch.equal(big.planes, plain.planes), f"max diff {(big.planes.float() - plain.planes.float()).abs().max().item():.3e}"
    _same(plain, ref_old, fused=case[7])
    for _ in range(3):
        _, again = _run_pair(*case, seed=sum(case[:5]) + 2, variant=3)
        assert torch.equal(again.planes, big.planes)


def test_tc_kernels_fuzz_bit_identical():
    """40 seeded random geometries: automatic variant choice (pipelined where instantiated, plain otherwise) against the plain
    kernel and against the packed output of the round-2 kernel."""
    rnd = random.Random(20260928)
    for case in range(40):
        K = rnd.choice([1, 2, 7, 7, 3])
        Cin = rnd.choice([32, 64, 96, 128, 256, 512])
        Cout = rnd.choice([32, 64, 96, 128, 256])
        dil = rnd.choice([1, 3, 5, 9]) if K == 7 else 1
        T = rnd.choice([37, 64, 333, 1000, 2049, 5000])
        B = rnd.choice([1, 2, 3])
        rep = rnd.choice([1, 1, 2]) if K != 2 else 1
        phases = rnd.choice([2, 4, 5]) if K == 2 else 1
        trim = rnd.choice([0, 1]) if phases > 1 else 0
        fused = K == 7 and Cin == Cout and Cin in (32, 64, 128) and rep == 1 and rnd.random() < 0.5
        residual = (not fused) and rnd.random() < 0.5
        args = (Cin, Cout, K, dil, T, B, residual, fused, rep, phases, trim, rnd.random() < 0.5)
        old, auto = _run_pair(*args, seed=case, variant=0)
        _, plain = _run_pair(*args, seed=case, variant=1)
        assert torch.equal(auto.planes, plain.planes), (case, args)
        _same(plain, old, fused)


def test_tc_rejects_what_it_cannot_serve():
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import Ua2Error
    x = ops.tc_pack(torch.randn(1, 32, 64).cuda())
    w = torch.randn(32, 32, 3).cuda()
    hi, lo = ops.pack_conv_weight_x3(w)
    with pytest.raises(Ua2Error, match="pipelined"):
        ops.conv1d_tc(x, hi, lo, 3, 32, pad_left=2, Tout=64, variant=2)          # K = 3 has no pipelined instantiation
    x17 = ops.TC(torch.zeros(2, 1, 64, 48, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(Ua2Error, match="multiple of 32"):
        ops.conv1d_tc(x17, hi, lo, 3, 32, pad_left=2, Tout=64)


@pytest.mark.parametrize("case,variant", [
    ((64, 64, 7, 9, 30000, 1, False, True, 1, 1, 0, False), 2), ((128, 128, 7, 5, 9000, 1, False, True, 1, 1, 0, False), 2),
    ((512, 512, 7, 3, 1500, 1, False, False, 1, 1, 0, False), 2), ((512, 512, 1, 1, 1500, 1, True, False, 1, 1, 0, False), 2),
    ((64, 64, 7, 9, 60000, 1, False, True, 1, 1, 0, False), 3), ((32, 32, 7, 7, 120000, 1, False, True, 1, 1, 0, False), 3)])
def test_lds_dma_kernels_are_repeatable_under_foreign_traffic(case, variant):
    """Race screen for the hand-counted LDS-DMA waits of the pipelined and the big-tile kernel (an early fragment read or a
    late window would show up as rare wrong tiles): 25 launches of the same problem, with unrelated traffic on another stream
    shifting the arrival times, give the plain kernel's bits every time."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import ACT_PRELU
    Cin, Cout, K, dil, T, B, residual, fused, rep, phases, trim, pca = case
    g = torch.Generator().manual_seed(1234 + Cin + K)
    x = ops.tc_pack(_representable(torch.randn(B, Cin, T, generator=g)).cuda())
    w = (torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5).cuda()
    hi, lo = ops.pack_conv_weight_x3(w)
    kw = dict(dilation=dil, pad_left=dil * (K - 1), Tout=T, bias=torch.randn(Cout, generator=g).cuda(), post_act=ACT_PRELU,
              post_alpha=torch.tensor([0.2]).cuda())
    if fused:
        w2 = (torch.randn(Cout, Cout, 1, generator=g) / Cout ** 0.5).cuda()
        kw["fused2"] = (*ops.pack_conv_weight_x3(ops.tc_w2_order(w2)), torch.randn(Cout, generator=g).cuda(), torch.tensor([0.3]).cuda())
    elif residual:
        kw["residual"] = ops.tc_pack(_representable(torch.randn(B, Cout, T, generator=g)).cuda())
    ref = ops.conv1d_tc(x, hi, lo, K, Cout, variant=1, **kw).planes
    torch.cuda.synchronize()
    noise = torch.randn(4096, 4096, device="cuda")
    side = torch.cuda.Stream()
    for i in range(25):
        with torch.cuda.stream(side):
            for _ in range(1 + i % 3):
                noise = noise * 1.0001 + 0.5
        got = ops.conv1d_tc(x, hi, lo, K, Cout, variant=variant, **kw).planes
        assert torch.equal(got, ref), f"launch {i}"
    torch.cuda.synchronize()
