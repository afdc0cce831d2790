"""Moshi / Mimi streaming-transformer family (the files north_star names), host side.

Mirror of the reference's llm_modules/transformer.py == tools/tokenizer/MimiCodec/model/modules/transformer.py
(StreamingTransformer :598-695, StreamingTransformerLayer :430-590, StreamingMultiheadAttention :293-426,
LayerScale :76-98, RMSNorm :49-66, multi_linear :155-179), llm_modules/rope.py (interleaved-pair RoPE,
:12-68) and llm_modules/gating.py (ActivationGating :24-52): same module tree and parameter names
(`layers.N.self_attn.in_proj_weight`, `.self_attn.out_proj.weight`, `.norm1.weight|alpha`, `.linear1.weight`,
`.gating.linear_in.weight`, `.layer_scale_1.scale`, ...), so a Moshi/Mimi state dict loads key for key.

Scope: what the two instances in the reference use — Mimi's codec transformer (layer_norm, GELU FFN, rope,
layer_scale, context 250; MimiCodec.py:54-58) and the depth transformer `mllm_model.py:114-143` would have
built (rms_norm_f32, silu gating, no positional embedding, weights_per_step).  `sin` positional
embeddings, cross-attention and non-causal attention are not built.

Arithmetic: the same five fused launches per layer as the live decoder (csrc/ua2_gemv.hip, ua2_linear.hip,
ua2_attn.hip) with this family's flavours selected by flags: LayerNorm / Moshi-RMS prologue, interleaved RoPE
from a host-built fp32 table, K/V appended to the paged cache, attention limited to the last `context`
positions, LayerScale folded into the residual epilogue, exact-erf GELU or silu-gating epilogue.

Semantics = the reference's WHOLE-SEQUENCE forward (`KVCacheResult.from_kv`: mask `0 <= delta < context`,
transformer.py:398-406), also when fed incrementally through `forward(x, offset=...)`.  The reference's
*streaming* mode loses one slot of its ring cache (effective window
capacity - 1 once the ring is full); that off-by-one is NOT reproduced — a conscious divergence, it is a
bug of the ring bookkeeping, not of the model.
"""
import math

import torch
import torch.nn as nn

from ...... import ops
from ......_lib import (EPI_GELU, EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, NORM_LAYERNORM, NORM_RMS_MOSHI,
                        PRO_CAST, PRO_NORM, ROPE_INTERLEAVED, ROPE_NONE, UA2_PAGE)

from .streaming import StreamingModule


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5, dtype=None, device=None):
        super().__init__()
        self.eps = eps
        self.alpha = nn.Parameter(torch.full((1, 1, dim), 1.0, device=device))


class LayerScale(nn.Module):
    def __init__(self, channels, init=1e-4, channel_last=True, device=None, dtype=None):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels,), init, device=device))


def create_norm_fn(norm_type, dim, device=None, **kw):
    if norm_type == "layer_norm":
        return nn.LayerNorm(dim, eps=1e-5, device=device)
    if norm_type == "rms_norm":
        return RMSNorm(dim, eps=1e-5, device=device)
    if norm_type == "rms_norm_f32":
        return RMSNorm(dim, eps=1e-8, device=device)
    raise ValueError(f"Unknown norm type: {norm_type}")


class ActivationGating(nn.Module):
    """gating.py:24-52 with activation = silu."""

    def __init__(self, dim, dim_feedforward, device=None):
        super().__init__()
        hidden = (21 * dim) // 8 if dim_feedforward == 4 * dim else (2 * dim_feedforward) // 3
        self.hidden = hidden
        self.linear_in = nn.Linear(dim, 2 * hidden, bias=False, device=device)
        self.linear_out = nn.Linear(hidden, dim, bias=False, device=device)


class StreamingMultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, causal=False, context=None, rope=None, weights_per_step=0, device=None):
        super().__init__()
        if not causal:
            raise NotImplementedError("only causal self-attention is on the hot path")
        self.embed_dim, self.num_heads, self.causal, self.context, self.rope = embed_dim, num_heads, causal, context, rope
        self.weights_per_step = weights_per_step
        mult = weights_per_step or 1
        self.in_proj_weight = nn.Parameter(torch.empty(mult * 3 * embed_dim, embed_dim, device=device))
        self.out_proj = nn.Linear(embed_dim, mult * embed_dim, bias=False, device=device)


class StreamingTransformerLayer(nn.Module):
    def __init__(self, d_model, num_heads, dim_feedforward=2048, causal=False, context=None, rope=None, norm="layer_norm",
                 layer_scale=None, gating="none", weights_per_step=0, device=None):
        super().__init__()
        self.self_attn = StreamingMultiheadAttention(d_model, num_heads, causal=causal, context=context, rope=rope,
                                                     weights_per_step=weights_per_step, device=device)
        self.norm1 = create_norm_fn(norm, d_model, device=device)
        self.norm2 = create_norm_fn(norm, d_model, device=device)
        self.weights_per_step = weights_per_step
        self.gating = self.linear1 = self.linear2 = None
        if gating == "none":
            assert not weights_per_step, "weights_per_step without gating not supported for now."
            self.linear1 = nn.Linear(d_model, dim_feedforward, bias=False, device=device)
            self.linear2 = nn.Linear(dim_feedforward, d_model, bias=False, device=device)
        elif gating == "silu":
            if weights_per_step:
                dims = dim_feedforward if isinstance(dim_feedforward, list) else [dim_feedforward] * weights_per_step
                self.gating = nn.ModuleList([ActivationGating(d_model, d, device=device) for d in dims])
            else:
                self.gating = ActivationGating(d_model, dim_feedforward, device=device)
        else:
            raise NotImplementedError(f"gating {gating!r}: only 'none' and 'silu' are on the hot path")
        if layer_scale is None:
            self.layer_scale_1, self.layer_scale_2 = nn.Identity(), nn.Identity()
        else:
            self.layer_scale_1 = LayerScale(d_model, layer_scale, device=device)
            self.layer_scale_2 = LayerScale(d_model, layer_scale, device=device)


class RotaryEmbedding(nn.Module):
    def __init__(self, max_period: float = 10000.0):
        super().__init__()
        self.max_period = max_period


def rope_tables(max_pos, head_dim, max_period):
    """cos/sin [max_pos, head_dim/2], fp32, with the ops of rope.py:39-56."""
    ds = torch.arange(head_dim // 2, dtype=torch.float32)
    freqs = torch.exp(ds * (-math.log(max_period) * 2 / head_dim))
    ts = torch.arange(max_pos, dtype=torch.float32).view(-1, 1)
    return torch.cos(freqs * ts), torch.sin(freqs * ts)


def create_sin_embedding(positions, dim, max_period=10000.0, dtype=torch.float32):
    """transformer.py:127-152: [cos(p / max_period^(i/(half-1))) , sin(...)], i < dim / 2.  Table arithmetic, kept in torch."""
    assert dim % 2 == 0
    half_dim = dim // 2
    positions = positions.to(dtype)
    adim = torch.arange(half_dim, device=positions.device, dtype=dtype).view(1, 1, -1)
    max_period_tensor = torch.full([], max_period, device=positions.device, dtype=dtype)
    phase = positions / (max_period_tensor ** (adim / (half_dim - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)


class StreamingTransformer(StreamingModule):
    def __init__(self, d_model, num_heads, num_layers, dim_feedforward=2048, causal=False, context=None,
                 positional_embedding="sin", max_period=10_000, positional_scale=1.0, betas=None, device=None, dtype=None,
                 **kwargs):
        super().__init__()
        assert d_model % num_heads == 0
        assert positional_embedding in {"sin", "rope", "sin_rope", "none"}
        self.d_model, self.num_heads, self.context = d_model, num_heads, context
        self.positional_embedding, self.max_period, self.positional_scale = positional_embedding, max_period, positional_scale
        self.rope = RotaryEmbedding(max_period) if positional_embedding in {"rope", "sin_rope"} else None
        self.layers = nn.ModuleList([StreamingTransformerLayer(d_model, num_heads, dim_feedforward, causal=causal, context=context,
                                                               rope=self.rope, device=device, **kwargs) for _ in range(num_layers)])
        self._plan = None
        self._plan_capacity = self._plan_batch = 0
        self.streaming_capacity = 2048          # positions one streaming session may span WITHOUT a context (linear cache)
        self.ring_chunk = 64                    # positions per launch of a ring-cached streaming session
        self.rope_capacity = 1 << 16            # positions the RoPE table of a ring-cached session covers (about 87 min at 12.5 Hz)

    def _init_streaming_state(self, batch_size: int):
        return {"offset": 0}

    # ---- device plan -----------------------------------------------------------------------------
    def prepare(self, max_batch=1, max_seq_length=1024, dtype=torch.float32, ring=False):
        """ring: streaming with a finite `context` (transformer.py:211-278 RingKVCache) — position p lives in page
        (p / 64) % ring_pages, so a session of any length re-uses ceil((context + ring_chunk) / 64) + 1 pages per sequence; the
        RoPE table is replaced by a long one (positions are unbounded).  The reference ring's off-by-one (the slot
        about to be overwritten is masked, so the streaming window is context - 1) is consciously NOT reproduced: streaming equals
        the non-streaming forward, which keeps `context` keys."""
        self._plan_capacity, self._plan_batch = max_seq_length, max_batch
        dev = self.layers[0].self_attn.in_proj_weight.device
        if dev.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback)")
        d, H = self.d_model, self.num_heads
        hs = d // H
        if hs % 16:
            raise NotImplementedError("head size must be a multiple of 16")
        pk = lambda w: ops.pack_linear(w.detach().float().contiguous(), dtype)
        f32 = lambda t: t.detach().float().reshape(-1).contiguous()
        plan = dict(dtype=dtype, dev=dev, hs=hs, layers=[])
        for L in self.layers:
            wps = L.weights_per_step or 1
            e = dict(wps=wps)
            e["qkv"] = [pk(w) for w in L.self_attn.in_proj_weight.view(wps, 3 * d, d)]
            e["out"] = [pk(w) for w in L.self_attn.out_proj.weight.view(wps, d, d)]
            for i, n in ((1, L.norm1), (2, L.norm2)):
                if isinstance(n, nn.LayerNorm):
                    e[f"n{i}"] = (NORM_LAYERNORM, f32(n.weight), f32(n.bias), n.eps)
                else:
                    e[f"n{i}"] = (NORM_RMS_MOSHI, f32(n.alpha), None, n.eps)
            e["ls1"] = f32(L.layer_scale_1.scale) if isinstance(L.layer_scale_1, LayerScale) else None
            e["ls2"] = f32(L.layer_scale_2.scale) if isinstance(L.layer_scale_2, LayerScale) else None
            if L.gating is None:
                e["ff"] = ("gelu", pk(L.linear1.weight), pk(L.linear2.weight), L.linear1.out_features)
            else:
                gs = list(L.gating) if isinstance(L.gating, nn.ModuleList) else [L.gating]
                e["ff"] = ("silu", [(pk(g.linear_in.weight[:g.hidden]), pk(g.linear_in.weight[g.hidden:]), pk(g.linear_out.weight), g.hidden)
                                    for g in gs])
            plan["layers"].append(e)
        n_pages = (max_seq_length + UA2_PAGE - 1) // UA2_PAGE
        plan["ring_pages"] = 0
        if ring:
            n_pages = (self.context + self.ring_chunk + UA2_PAGE - 1) // UA2_PAGE + 1
            n_pages = 1 << (n_pages - 1).bit_length()             # the kernels index the ring with a mask (include/ua2hip.h)
            plan["ring_pages"] = n_pages
        plan["max_pages"] = n_pages
        shape = (max_batch * n_pages, H, UA2_PAGE, hs)
        plan["k"] = [torch.zeros(shape, dtype=dtype, device=dev) for _ in self.layers]
        plan["v"] = [torch.zeros(shape, dtype=dtype, device=dev) for _ in self.layers]
        plan["ptab"] = torch.arange(max_batch * n_pages, dtype=torch.int32, device=dev).view(max_batch, n_pages)
        if self.rope is not None:
            cos, sin = rope_tables(max_seq_length, hs, self.max_period)
            plan["cos"], plan["sin"] = cos.to(dev).contiguous(), sin.to(dev).contiguous()
        self._plan = plan
        return self

    def _run_rows(self, xs, row_pos, row_seq, step):
        """xs (R, d) fp32 in place; all rows use the weights of `step` (0 without weights_per_step)."""
        p, d, H = self._plan, self.d_model, self.num_heads
        dt, hs, dev, R = p["dtype"], p["hs"], xs.device, xs.shape[0]
        q = torch.empty(R, d, dtype=torch.float32, device=dev)
        y = torch.empty(R, d, dtype=torch.float32, device=dev)
        rope_mode = ROPE_INTERLEAVED if self.rope is not None else ROPE_NONE
        for li, e in enumerate(p["layers"]):
            s = step if e["wps"] > 1 else 0
            kind, w, b, eps = e["n1"]
            geom = ops.kv_geom(p["k"][li], p["v"][li], p["ptab"], H, H, hs, ring_pages=p["ring_pages"])
            ops.linear(dtype=dt, M=R, N=3 * d, K=d, w0=e["qkv"][s], prologue=PRO_NORM, epilogue=EPI_QKV_ROPE, x=xs, norm_w=w,
                       norm_b=b, norm_kind=kind, eps=eps, row_pos=row_pos, row_seq=row_seq, rope_cos=p.get("cos"),
                       rope_sin=p.get("sin"), rope_mode=rope_mode, q_out=q, kv=geom)
            ops.attn(dtype=dt, R=R, q=q, row_pos=row_pos, row_seq=row_seq, kv=geom, y=y, window=self.context or 0)
            ops.linear(dtype=dt, M=R, N=d, K=d, w0=e["out"][s], prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=y, y=xs, resid=xs,
                       out_scale=e["ls1"])
            kind, w, b, eps = e["n2"]
            if e["ff"][0] == "gelu":
                _, w1, w2, F = e["ff"]
                act = torch.empty(R, F, dtype=torch.float32, device=dev)
                ops.linear(dtype=dt, M=R, N=F, K=d, w0=w1, prologue=PRO_NORM, epilogue=EPI_GELU, x=xs, norm_w=w, norm_b=b,
                           norm_kind=kind, eps=eps, y=act)
            else:
                wi0, wi1, w2, F = e["ff"][1][s if len(e["ff"][1]) > 1 else 0]
                act = torch.empty(R, F, dtype=torch.float32, device=dev)
                ops.linear(dtype=dt, M=R, N=F, K=d, w0=wi0, w1=wi1, prologue=PRO_NORM, epilogue=EPI_SWIGLU, x=xs, norm_w=w,
                           norm_b=b, norm_kind=kind, eps=eps, y=act)
            ops.linear(dtype=dt, M=R, N=d, K=F, w0=w2, prologue=PRO_CAST, epilogue=EPI_RESIDUAL, x=act, y=xs, resid=xs,
                       out_scale=e["ls2"])

    @torch.inference_mode()
    def forward(self, x, offset: int = 0):
        """x (B, T, C) at positions offset..offset+T-1 (offset = 0: whole sequence; > 0: incremental decoding
        against the cached keys).  With weights_per_step, position t uses the weights of step t (multi_linear)."""
        state = self._streaming_state
        if state is not None:                     # streaming(): positions continue where the previous call stopped
            assert offset == 0, "pass either an explicit offset or use streaming(), not both"
            offset = state["offset"]
            state["offset"] = offset + x.shape[1]
            if self.context:                                  # finite context: ring cache, unbounded session length
                if self._plan is None or not self._plan["ring_pages"] or self._plan_batch < x.shape[0]:
                    self.prepare(max_batch=x.shape[0], max_seq_length=self.rope_capacity, ring=True,
                                 **(dict(dtype=self._plan["dtype"]) if self._plan is not None else {}))
                if offset + x.shape[1] > self.rope_capacity and self.rope is not None:
                    raise RuntimeError(f"the RoPE table covers {self.rope_capacity} positions; raise rope_capacity for longer sessions")
                if x.shape[1] > self.ring_chunk:       # keep (context + positions per launch) inside the ring
                    state["offset"] = offset
                    return torch.cat([self.forward(x[:, t:t + self.ring_chunk]) for t in range(0, x.shape[1], self.ring_chunk)], dim=1)
            else:
                if offset + x.shape[1] > self.streaming_capacity:
                    raise RuntimeError(f"streaming past {self.streaming_capacity} positions without a context (an unbounded cache)")
                if self._plan is None or self._plan_capacity < self.streaming_capacity or self._plan_batch < x.shape[0]:
                    self.prepare(max_batch=x.shape[0], max_seq_length=self.streaming_capacity,
                                 **(dict(dtype=self._plan["dtype"]) if self._plan is not None else {}))
        if state is None:
            # Whole-sequence / explicit-offset calls need a LINEAR cache that covers every position of the launch.  A plan left
            # behind by a streaming session with a finite context is a ring of a few pages: the QKV epilogue writes all T rows
            # before attention runs, so T > ring span would overwrite slots that earlier rows of the same launch still attend to
            # (silently wrong output).  Never reuse it here; and never run a linear plan past its capacity or batch.
            need = max(UA2_PAGE, offset + x.shape[1])
            stale = self._plan is None or self._plan["ring_pages"] > 0 or self._plan_capacity < need or self._plan_batch < x.shape[0]
            if stale:
                if offset > 0:
                    raise RuntimeError("forward(offset > 0) continues a cache this plan does not hold (ring plan of a streaming session, or "
                                       f"capacity {self._plan_capacity} < {need}): call prepare(max_batch, max_seq_length) before the first chunk")
                kw = dict(dtype=self._plan["dtype"]) if self._plan is not None else {}      # keep the numerics contract the caller chose
                self.prepare(max_batch=x.shape[0], max_seq_length=need, **kw)
        elif self._plan["ring_pages"] > 0:
            assert x.shape[1] <= self.ring_chunk, "a ring-cached launch may span at most ring_chunk positions"
        B, T, Cc = x.shape
        dev = x.device
        out = x.float().contiguous().clone()
        if self.positional_embedding in {"sin", "sin_rope"}:        # transformer.py:683-689
            positions = (offset + torch.arange(T, device=dev)).view(1, -1, 1)
            out = out + self.positional_scale * create_sin_embedding(positions, Cc, max_period=self.max_period)
        if any(e["wps"] > 1 for e in self._plan["layers"]):
            seq = torch.arange(B, dtype=torch.int32, device=dev)
            for t in range(T):
                xs = out[:, t].contiguous()
                self._run_rows(xs, torch.full((B,), offset + t, dtype=torch.int32, device=dev), seq, offset + t)
                out[:, t] = xs
            return out
        xs = out.view(B * T, Cc)
        pos = (offset + torch.arange(T, device=dev, dtype=torch.int32)).repeat(B)
        seq = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(T)
        self._run_rows(xs, pos.contiguous(), seq.contiguous(), 0)
        return xs.view(B, T, Cc)


class ProjectedTransformer(nn.Module):
    """transformer.py:698-750: optional Linear in / out projections around a StreamingTransformer, (B, C, T) layout."""

    def __init__(self, input_dimension, output_dimensions, d_model, *, conv_layout=False, **kwargs):
        super().__init__()
        self.transformer = StreamingTransformer(d_model=d_model, **kwargs)
        self.input_dimension, self.output_dimensions, self.conv_layout = input_dimension, output_dimensions, conv_layout
        self.input_proj = nn.Linear(input_dimension, d_model, bias=False) if d_model != input_dimension else None
        self.output_projs = nn.ModuleList([nn.Identity() if d_model == od else nn.Linear(d_model, od, bias=False)
                                           for od in output_dimensions])

    def _proj(self, lin, x2d):
        w = ops.pack_linear(lin.weight.detach().float(), torch.float32)
        y = torch.empty(x2d.shape[0], lin.out_features, dtype=torch.float32, device=x2d.device)
        ops.linear(dtype=torch.float32, M=x2d.shape[0], N=lin.out_features, K=lin.in_features, w0=w, prologue=PRO_CAST,
                   epilogue=EPI_STORE, x=x2d, y=y)
        return y

    @torch.inference_mode()
    def forward(self, x):
        if self.conv_layout:
            x = x.transpose(1, 2)
        B, T, _ = x.shape
        x = x.float().contiguous()
        if self.input_proj is not None:
            x = self._proj(self.input_proj, x.view(B * T, -1)).view(B, T, -1)
        z = self.transformer(x)
        ys = []
        for op in self.output_projs:
            y = z if isinstance(op, nn.Identity) else self._proj(op, z.reshape(B * T, -1).contiguous()).view(B, T, -1)
            ys.append(y.transpose(1, 2) if self.conv_layout else y)
        return ys
