"""Row invariance of the linear operator at the model's real shapes: a row's output must not depend on how many
other rows share the launch (rows per workgroup and launch geometry are functions of N, K and the
dtype only; per-row statistics are reduced in one fixed order).  Bit-exact comparisons of M = 1 launches against
the same rows inside M = 2, 5, 16, 17 and 40-row launches."""
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
