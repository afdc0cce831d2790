"""Residual vector quantizers with input / output projections, host side (inference subset).

Mirror of tools/tokenizer/MimiCodec/model/quantization/vq.py: ResidualVectorQuantizer :19-172 (1x1 Conv1d
input_proj / output_proj without bias :79-90, encode :134-147, decode :149-155) and
SplitResidualVectorQuantizer :174-358 (first `n_q_semantic` levels and the rest quantise the SAME input through
their own projections :305-315; decode adds the two de-quantised halves :317-324).  Same attribute tree =>
same state-dict keys.  forward() (training: dropout over levels, commitment loss, semantic distillation) is
out of scope.
"""
import torch
import torch.nn as nn

from ...... import ops
from .core_vq import ResidualVectorQuantization


class _Proj1x1(nn.Conv1d):
    """nn.Conv1d(k=1, bias=False) whose forward is ua2_conv1d (exact fp32)."""

    def forward(self, x):
        if getattr(self, "_packed", None) is None:
            self._packed = ops.pack_conv_weight(self.weight.detach().float())
        w, k = self._packed
        return ops.conv1d(x.float().contiguous(), w, k, self.out_channels, Tout=x.shape[-1])


class ResidualVectorQuantizer(nn.Module):
    def __init__(self, dimension=128, input_dimension=None, output_dimension=None, n_q=8, q_dropout=False,
                 q_first_only_proba=0.0, no_quantization_rate=0.0, bins=1024, decay=0.99, threshold_usage_ratio=0.1,
                 replaced_usage_ratio=1.0, codebook_offset=0, force_projection=False, generator_seed=None):
        super().__init__()
        self.max_n_q = self.n_q = n_q
        self.dimension = dimension
        self.input_dimension = input_dimension or dimension
        self.output_dimension = output_dimension or dimension
        self.bins = bins
        if self.input_dimension == self.dimension and not force_projection:
            self.input_proj = nn.Identity()
        else:
            self.input_proj = _Proj1x1(self.input_dimension, self.dimension, 1, bias=False)
        if self.output_dimension == self.dimension and not force_projection:
            self.output_proj = nn.Identity()
        else:
            self.output_proj = _Proj1x1(self.dimension, self.output_dimension, 1, bias=False)
        self.vq = ResidualVectorQuantization(dim=self.dimension, codebook_size=self.bins, num_quantizers=self.n_q, decay=decay,
                                             threshold_usage_ratio=threshold_usage_ratio,
                                             replaced_usage_ratio=replaced_usage_ratio, codebook_offset=codebook_offset)

    @torch.inference_mode()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, input_dimension, T) -> codes (B, K, T) int64 (vq.py:134-147)."""
        if x.shape[-1] == 0:
            return torch.empty((x.shape[0], self.n_q, 0), device=x.device, dtype=torch.int64)
        return self.vq.encode(self.input_proj(x), n_q=self.n_q).transpose(0, 1)

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (B, K, T) -> (B, output_dimension, T) (vq.py:149-155)."""
        return self.output_proj(self.vq.decode(codes.transpose(0, 1)))

    def forward(self, *a, **k):
        raise NotImplementedError("training-time forward (quantizer dropout, commitment loss) is out of scope; use encode/decode")

    @property
    def total_codebooks(self):
        return self.max_n_q

    @property
    def num_codebooks(self):
        return self.n_q

    def set_num_codebooks(self, n: int):
        assert 0 <= n <= self.max_n_q
        self.n_q = n

    @property
    def cardinality(self) -> int:
        return self.bins


class SplitResidualVectorQuantizer(nn.Module):
    def __init__(self, *, n_q=8, no_quantization_rate=0.0, no_quantization_mode="same", n_q_semantic=1, **kwargs):
        super().__init__()
        assert n_q > n_q_semantic, f"Number of quantizers {n_q} must be larger than the number of semantic quantizers {n_q_semantic}."
        self.max_n_q = n_q
        self.n_q_semantic, self.n_q_acoustic = n_q_semantic, n_q - n_q_semantic
        kwargs.pop("q_dropout", None)
        kwargs.pop("generator_seed", None)
        self.rvq_first = ResidualVectorQuantizer(n_q=n_q_semantic, force_projection=True, q_dropout=False, **kwargs)
        self.rvq_rest = ResidualVectorQuantizer(n_q=n_q - n_q_semantic, codebook_offset=1, force_projection=True, **kwargs)

    @torch.inference_mode()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """vq.py:305-315: both halves quantise x itself; codes (B, K, T)."""
        codes = self.rvq_first.encode(x)
        if self.n_q > self.n_q_semantic:
            codes = torch.cat([codes, self.rvq_rest.encode(x)], dim=1)
        return codes

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """vq.py:317-324."""
        quantized = self.rvq_first.decode(codes[:, : self.n_q_semantic])
        if codes.shape[1] > self.n_q_semantic:
            quantized = quantized + self.rvq_rest.decode(codes[:, self.n_q_semantic:])
        return quantized

    def forward(self, *a, **k):
        raise NotImplementedError("training-time forward (semantic distillation loss) is out of scope; use encode/decode")

    @property
    def total_codebooks(self):
        return self.rvq_first.max_n_q + self.rvq_rest.max_n_q

    @property
    def num_codebooks(self):
        return self.rvq_first.num_codebooks + self.rvq_rest.num_codebooks

    @property
    def n_q(self):
        return self.rvq_first.n_q + self.rvq_rest.n_q

    @property
    def dimension(self):
        return self.rvq_first.dimension

    @property
    def semantic_quantizer(self) -> ResidualVectorQuantizer:
        return self.rvq_first

    @property
    def acoustic_quantizer(self) -> ResidualVectorQuantizer:
        return self.rvq_rest

    def set_num_codebooks(self, n: int):
        assert self.n_q_semantic <= n <= self.total_codebooks
        self.rvq_rest.set_num_codebooks(n - self.n_q_semantic)

    @property
    def cardinality(self) -> int:
        assert self.rvq_rest.cardinality == self.rvq_first.cardinality
        return self.rvq_first.cardinality
