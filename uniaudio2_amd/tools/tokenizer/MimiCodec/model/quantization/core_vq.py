"""Euclidean-codebook residual VQ of the Mimi family, host side (inference subset).

Mirror of tools/tokenizer/MimiCodec/model/quantization/core_vq.py: EuclideanCodebook :73-220 (buffers
`_initialized`, `cluster_usage`, `embedding_sum`; `embedding` = embedding_sum / clamp(cluster_usage, eps) :143-149;
`_quantize` = cdist -> argmin :179-185), VectorQuantization :222-308 (optional project_in / project_out),
ResidualVectorQuantization.encode :365-376 / .decode :378-384.  Same attribute tree => same state-dict keys.
The search and the lookup run in ua2_rvq_encode / ua2_rvq_decode (bit-exact contract of oracle/rvq_oracle.c:
squared L2 as an ascending fma chain, lowest index on ties); training-time code (EMA updates, dead-code
replacement, straight-through) is out of scope.
"""
import torch
import torch.nn as nn

from ...... import ops


class EuclideanCodebook(nn.Module):
    def __init__(self, dim, codebook_size, decay=0.99, epsilon=1e-5, threshold_usage_ratio=0.1, replaced_usage_ratio=1.0,
                 check_unused_every=5):
        super().__init__()
        self.dim, self.codebook_size, self.epsilon = dim, codebook_size, epsilon
        self.register_buffer("_initialized", torch.tensor([False], dtype=torch.float))
        self.register_buffer("cluster_usage", torch.ones(codebook_size))
        self.register_buffer("embedding_sum", torch.zeros(codebook_size, dim))

    @property
    def embedding(self) -> torch.Tensor:
        return self.embedding_sum / self.cluster_usage.clamp(min=self.epsilon)[:, None]           # core_vq.py:143-149


class VectorQuantization(nn.Module):
    def __init__(self, dim, codebook_size, codebook_dim=None, decay=0.99, epsilon=1e-5, threshold_usage_ratio=0.1, **kwargs):
        super().__init__()
        codebook_dim = codebook_dim or dim
        if codebook_dim != dim:
            raise NotImplementedError("per-level project_in / project_out (codebook_dim != dim) is not used by MimiCodec")
        self.project_in, self.project_out = nn.Identity(), nn.Identity()
        self.epsilon = epsilon
        self._codebook = EuclideanCodebook(dim=codebook_dim, codebook_size=codebook_size, decay=decay, epsilon=epsilon,
                                           threshold_usage_ratio=threshold_usage_ratio, **kwargs)
        self.codebook_size = codebook_size

    @property
    def embedding(self):
        return self._codebook.embedding


class ResidualVectorQuantization(nn.Module):
    """core_vq.py:311-384.  `codebook_offset` only affects training-time metrics names."""

    def __init__(self, *, num_quantizers: int, codebook_offset: int = 0, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([VectorQuantization(**kwargs) for _ in range(num_quantizers)])
        self.codebook_offset = codebook_offset
        self._emb = None

    def _tables(self):
        if self._emb is None:
            emb = torch.stack([l.embedding for l in self.layers]).float().contiguous()           # [L, C, D]
            if emb.device.type != "cuda":
                raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback)")
            self._emb = (emb, emb.transpose(1, 2).contiguous())
        return self._emb

    @torch.inference_mode()
    def encode(self, x: torch.Tensor, n_q=None) -> torch.Tensor:
        """x (B, D, T) -> codes (K, B, T) int64: level k quantises what levels < k left over (:365-376)."""
        emb, embT = self._tables()
        n_q = n_q or len(self.layers)
        B, D, T = x.shape
        rows = x.transpose(1, 2).reshape(B * T, D).float().contiguous()
        codes, _ = ops.rvq_encode(rows, emb[:n_q].contiguous(), embT[:n_q].contiguous())
        return codes.view(B, T, n_q).permute(2, 0, 1).long().contiguous()

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (K, B, T) -> (B, D, T): sum over levels of the selected codewords (:378-384)."""
        emb, _ = self._tables()
        K, B, T = codes.shape
        q = ops.rvq_decode(codes.permute(1, 2, 0).reshape(B * T, K).to(torch.int32).contiguous(), emb[:K].contiguous())
        return q.view(B, T, -1).transpose(1, 2).contiguous()
