"""Transformer block of the codec's AudioThinking encoder, host side.

Mirror of the reference's tools/tokenizer/ReasoningCodec_film/modules/transformer.py for the configuration
AudioDiffusion1D.py:177-179 instantiates (`TransformerBlock(dim, dim_heads=128, causal=False, power_normalized=True,
layer_scale=True, add_rope=True, attn_kwargs={'qk_norm': True}, ff_kwargs={'mult': 4, 'no_bias': False})`): same class
and attribute names, hence the same state-dict keys (`self_attn.to_qkv.parametrizations.weight.original0/1`,
`self_attn.q_norm.weight`, `self_attn_scale.scale`, `ff.ff.0.proj.bias`, `rope.inv_freq`, ...).  Power normalisation
removes the pre / ff norms (:672-674) and weight-norms every Linear (:218, 331-343, 302).

The modules only hold parameters.  `prepare()` folds the weight norm and packs the filters; `run()` issues per layer:
to_qkv GEMM -> ua2_qknorm_rope_kv (q/k LayerNorm over the head dim :452-455, rotary on the first max(dim_heads//2, 32) dims
:457-478, K/V to the paged layout) -> dense attention (:563-588) -> to_out GEMM with LayerScale + residual fused (:773)
-> GLU GEMM pair with x * sigmoid(gate) fused (:208-243) -> linear_out GEMM with LayerScale + residual fused (:781).
"""
import torch
import torch.nn as nn
from torch.nn.utils.parametrizations import weight_norm

from ..... import ops
from ....._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, GATE_SIGMOID_SECOND
from ..models._dense import DenseKV, PackedLinear, folded_weight


class LayerScale(nn.Module):
    def __init__(self, dim, init_val=1e-2):
        super().__init__()
        self.scale = nn.Parameter(torch.full([dim], init_val))


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, base=10000):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)))


class GLU(nn.Module):
    def __init__(self, dim_in, dim_out, no_bias=False):
        super().__init__()
        self.proj = weight_norm(nn.Linear(dim_in, dim_out * 2, bias=not no_bias))


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, no_bias=False):
        super().__init__()
        inner = int(dim * mult)
        self.ff = nn.Sequential(GLU(dim, inner, no_bias=no_bias), nn.Identity(), weight_norm(nn.Linear(inner, dim, bias=not no_bias), name="weight"),
                                nn.Identity())


class Attention(nn.Module):
    def __init__(self, dim, dim_heads=64, qk_norm=False):
        super().__init__()
        self.dim, self.dim_heads, self.num_heads = dim, dim_heads, dim // dim_heads
        self.to_qkv = weight_norm(nn.Linear(dim, dim * 3, bias=False), name="weight")
        self.to_out = weight_norm(nn.Linear(dim, dim, bias=False), name="weight")
        self.qk_norm = qk_norm
        if qk_norm:
            self.q_norm = nn.LayerNorm(dim_heads)
            self.k_norm = nn.LayerNorm(dim_heads)


class TransformerBlock(nn.Module):
    def __init__(self, dim, dim_heads=64, causal=False, power_normalized=True, layer_scale=True, add_rope=True, attn_kwargs=None,
                 ff_kwargs=None, **unused):
        super().__init__()
        if causal or not power_normalized or not layer_scale or not add_rope:
            raise NotImplementedError("only the AudioThinking configuration (non-causal, power-normalised, LayerScale, rotary) is built")
        self.dim, self.dim_heads = dim, dim_heads
        self.self_attn = Attention(dim, dim_heads=dim_heads, **(attn_kwargs or {}))
        self.self_attn_scale = LayerScale(dim)
        self.ff = FeedForward(dim, **(ff_kwargs or {}))
        self.ff_scale = LayerScale(dim)
        self.rope = RotaryEmbedding(max(dim_heads // 2, 32))
        self._p = None

    def prepare(self, dtype=torch.float32):
        f32 = lambda t: t.detach().float().contiguous()
        a = self.self_attn
        inner = self.ff.ff[2].in_features
        wg = folded_weight(self.ff.ff[0].proj)                     # rows [0, inner) = x, [inner, 2 inner) = gate (:236 chunk)
        bg = self.ff.ff[0].proj.bias
        self._p = dict(
            dtype=dtype, qkv=PackedLinear(folded_weight(a.to_qkv), None, dtype), out=PackedLinear(folded_weight(a.to_out), None, dtype),
            glu_x=PackedLinear(wg[:inner], bg[:inner] if bg is not None else None, dtype),
            glu_g=PackedLinear(wg[inner:], bg[inner:] if bg is not None else None, dtype),
            ff_out=PackedLinear(folded_weight(self.ff.ff[2]), self.ff.ff[2].bias, dtype),
            attn_scale=f32(self.self_attn_scale.scale), ff_scale=f32(self.ff_scale.scale), inv_freq=f32(self.rope.inv_freq),
            qn=(f32(a.q_norm.weight), f32(a.q_norm.bias), f32(a.k_norm.weight), f32(a.k_norm.bias)) if a.qk_norm else None)
        return self

    def rope_tables(self, T, device):
        """cos / sin of RotaryEmbedding.forward_from_seq_len (:120-134) — [T, rot_dim / 2] (the two halves of `freqs` are equal)."""
        f = torch.einsum("i,j->ij", torch.arange(T, device=device).float(), self._p["inv_freq"])
        return f.cos().contiguous(), f.sin().contiguous()

    def run(self, x, kv: DenseKV, cos, sin):
        """x [B*T, dim] fp32 rows (updated in place and returned)."""
        p = self._p
        qkv = p["qkv"](x)
        q = torch.empty(x.shape[0], self.dim, dtype=torch.float32, device=x.device)
        qn = p["qn"] or (None, None, None, None)
        ops.qknorm_rope_kv(p["dtype"], qkv, kv.row_pos, kv.row_seq, kv.geom, q, qw=qn[0], qb=qn[1], kw=qn[2], kb=qn[3], eps=1e-5,
                           cos=cos, sin=sin, rot_dim=2 * p["inv_freq"].numel())
        o = kv.attend(q)
        p["out"](o, epilogue=EPI_RESIDUAL, resid=x, out_scale=p["attn_scale"], y=x)
        g = p["glu_x"](x, epilogue=EPI_SWIGLU, w1=p["glu_g"], act_kind=GATE_SIGMOID_SECOND)
        p["ff_out"](g, epilogue=EPI_RESIDUAL, resid=x, out_scale=p["ff_scale"], y=x)
        return x
