"""Residual VQ of the live codec, host side (inference subset).

Stands in for `vector_quantize_pytorch.ResidualVQ` (pinned 1.27.15 in the reference's pyproject.toml:31,
NOT vendored and not installed here — parity with the real package is UNPINNED) as the
reference uses it in eval mode: ReasoningCodec_film/models/AudioDiffusion1D.py:183-187,256-264
(construction), :388,529,535,544 (`rvq(x)` -> quantized, indices, loss), :577-583
(`get_output_from_indices`).  Semantics restated from the package's documented algorithm and
cross-checked against the reference's vendored core_vq (same search / lookup, different projections):
  project_in (Linear dim -> codebook_dim, shared by all levels) -> per level nearest codeword by
  squared L2 (lowest index on ties), residual -= codeword -> sum of codewords -> project_out.
State-dict keys follow the package: project_in.weight/bias, project_out.weight/bias,
layers.{i}._codebook.embed ([1, codebook_size, codebook_dim]).
"""
import torch
import torch.nn as nn

from ..... import ops
from ....._lib import EPI_RESIDUAL, PRO_CAST


class _Codebook(nn.Module):
    def __init__(self, codebook_size, codebook_dim):
        super().__init__()
        self.register_buffer("embed", torch.zeros(1, codebook_size, codebook_dim))


class _Layer(nn.Module):
    def __init__(self, codebook_size, codebook_dim):
        super().__init__()
        self._codebook = _Codebook(codebook_size, codebook_dim)


class ResidualVQ(nn.Module):
    def __init__(self, *, dim, codebook_size, num_quantizers, codebook_dim=None, **unused_training_kwargs):
        super().__init__()
        codebook_dim = codebook_dim or dim
        self.dim, self.codebook_dim, self.codebook_size, self.num_quantizers = dim, codebook_dim, codebook_size, num_quantizers
        proj = codebook_dim != dim
        self.project_in = nn.Linear(dim, codebook_dim) if proj else nn.Identity()
        self.project_out = nn.Linear(codebook_dim, dim) if proj else nn.Identity()
        self.layers = nn.ModuleList([_Layer(codebook_size, codebook_dim) for _ in range(num_quantizers)])
        self._plan = None

    def _prepare(self):
        emb = torch.cat([l._codebook.embed for l in self.layers], 0).float().contiguous()     # [L, C, D]
        if emb.device.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback)")
        p = dict(emb=emb, embT=emb.transpose(1, 2).contiguous())
        for name in ("project_in", "project_out"):
            lin = getattr(self, name)
            if isinstance(lin, nn.Linear):
                p[name] = (ops.pack_linear(lin.weight.detach().float(), torch.float32), lin.bias.detach().float().contiguous(),
                           lin.out_features, lin.in_features)
        self._plan = p
        return p

    def _linear(self, key, x):
        """nn.Linear with bias on the exact-fp32 MFMA path: the bias rides in the residual slot (ldr = 0)."""
        p = self._plan
        if key not in p:
            return x
        w, b, N, K = p[key]
        y = torch.empty(x.shape[0], N, dtype=torch.float32, device=x.device)
        ops.linear(dtype=torch.float32, M=x.shape[0], N=N, K=K, w0=w, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=x, y=y,
                   resid=b, ldr=0)
        return y

    @torch.inference_mode()
    def forward(self, x):
        """x (B, T, dim) -> quantized (B, T, dim), indices (B, T, L) int64, commit loss (zeros) — eval mode."""
        p = self._plan or self._prepare()
        B, T, _ = x.shape
        h = self._linear("project_in", x.reshape(B * T, -1).float().contiguous())
        codes, q = ops.rvq_encode(h, p["emb"], p["embT"])
        out = self._linear("project_out", q)
        return out.view(B, T, -1), codes.view(B, T, -1).long(), torch.zeros(1, self.num_quantizers, device=x.device)

    @torch.inference_mode()
    def get_output_from_indices(self, indices):
        """indices (B, T, L) -> (B, T, dim): sum of the selected codewords, then project_out."""
        p = self._plan or self._prepare()
        B, T, L = indices.shape
        q = ops.rvq_decode(indices.reshape(B * T, L).to(torch.int32).contiguous(), p["emb"])
        return self._linear("project_out", q).view(B, T, -1)
