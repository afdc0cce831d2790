"""GPU parity of the conv kernel against the PyTorch fp32 ops the reference calls (F.conv1d /
F.conv_transpose1d on CPU), tolerance 2e-5 relative to the output scale (fp32, different summation order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=2e-5):
    scale = float(ref.abs().max()) + 1e-6
    err = float((got.cpu() - ref).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


CONV_CASES = [
    # Cin, Cout, K, stride, dil, causal, T
    (1, 32, 7, 1, 1, True, 1000), (32, 32, 7, 1, 9, True, 777), (48, 48, 7, 1, 5, False, 300), (64, 64, 1, 1, 1, True, 130),
    (32, 64, 4, 2, 1, True, 1001), (136, 512, 7, 1, 1, False, 100), (64, 128, 10, 5, 1, True, 640), (1024, 768, 2, 2, 1, False, 64),
    (64, 128, 16, 8, 1, True, 803), (128, 64, 3, 1, 2, True, 97), (17, 19, 5, 1, 3, False, 70),
]


@pytest.mark.parametrize("Cin,Cout,K,stride,dil,causal,T", CONV_CASES)
def test_conv1d_matches_torch(Cin, Cout, K, stride, dil, causal, T):
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import ACT_PRELU
    g = torch.Generator().manual_seed(Cin * 7 + K)
    x = torch.randn(2, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    alpha = torch.tensor([0.2])
    if causal:                                   # scalar24k.py:70-74 (stride 1) / conv.py:239-243 (left pad k_eff - stride, extra right pad)
        pad_l = dil * (K - 1) - (stride - 1)
        n_frames = -(-(T - (dil * (K - 1) + 1) + pad_l) // stride) + 1
        extra = (n_frames - 1) * stride + (dil * (K - 1) + 1) - pad_l - T
        xp = F.pad(x, (pad_l, max(extra, 0)))
    else:                                        # scalar24k.py:17,47 symmetric
        pad_l = (K * dil - dil) // 2
        xp = F.pad(x, (pad_l, pad_l))
    ref = F.prelu(F.conv1d(xp, w, b, stride=stride, dilation=dil), alpha)
    res = torch.randn(ref.shape, generator=g)
    ref = ref + res
    wp, _ = ops.pack_conv_weight(w.cuda())
    y = ops.conv1d(x.cuda(), wp, K, Cout, stride=stride, dilation=dil, pad_left=pad_l, Tout=ref.shape[-1], bias=b.cuda(),
                   post_act=ACT_PRELU, post_alpha=alpha.cuda(), residual=res.cuda())
    _close(y, ref)


@pytest.mark.parametrize("Cin,Cout,stride,K,causal,T", [(64, 32, 2, 4, True, 333), (128, 64, 5, 10, True, 64), (32, 32, 3, 6, True, 200),
                                                        (64, 64, 4, 8, False, 100), (512, 256, 8, 16, True, 25), (16, 16, 2, 2, False, 50)])
def test_conv_transpose1d_matches_torch(Cin, Cout, stride, K, causal, T):
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(Cin + stride)
    x = torch.randn(2, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, K, generator=g) / (Cin * K / stride) ** 0.5
    b = torch.randn(Cout, generator=g)
    if causal:                                   # scalar24k.py:91-93,108-112: padding 0, drop the last `stride` samples
        ref = F.conv_transpose1d(x, w, b, stride=stride)[:, :, :-stride]
        trim = 0
    else:                                        # padding = (k - s) // 2
        trim = (K - stride) // 2
        ref = F.conv_transpose1d(x, w, b, stride=stride, padding=trim)
    wp, M = ops.pack_convtr_weight(w.cuda(), stride)
    y = ops.conv1d(x.cuda(), wp, M, Cout, pad_left=M - 1, Tout=ref.shape[-1], bias=b.cuda(), out_phases=stride, out_trim_left=trim)
    _close(y, ref)


def test_conv1d_pre_activation_repeat_and_pool():
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import ACT_ELU, ACT_ROUND9, ACT_TANH
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 150, generator=g)
    w = torch.randn(64, 64, 3, generator=g) / 14
    wp, _ = ops.pack_conv_weight(w.cuda())
    # SEANet resblock step: ELU -> conv(k=3, dilation 2), causal (seanet.py:92-94)
    ref = F.conv1d(F.pad(F.elu(x), (4, 0)), w, dilation=2)
    _close(ops.conv1d(x.cuda(), wp, 3, 64, dilation=2, pad_left=4, Tout=150, pre_act=ACT_ELU), ref)
    # PostProcessor: repeat each step twice, then conv (scalar24k.py:136-140); tanh epilogue (scalar24k.py:385)
    xr = x.transpose(1, 2).repeat(1, 1, 2).view(1, -1, 64).transpose(1, 2)
    ref = torch.tanh(F.conv1d(F.pad(xr, (2, 0)), w))
    _close(ops.conv1d(x.cuda(), wp, 3, 64, pad_left=2, Tout=300, in_repeat=2, post_act=ACT_TANH), ref)
    # decode entry: round(9x)/9 on the input (scalar24k.py:404)
    ref = F.conv1d(F.pad(torch.round(9 * x) / 9, (1, 1)), w)
    _close(ops.conv1d(x.cuda(), wp, 3, 64, pad_left=1, Tout=150, pre_act=ACT_ROUND9), ref)
    _close(ops.avgpool1d(x.cuda(), 2), F.avg_pool1d(x, 2), tol=1e-6)
