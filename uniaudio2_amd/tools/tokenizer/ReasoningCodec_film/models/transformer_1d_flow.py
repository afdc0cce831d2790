"""Flow-matching DiT of the codec's decoder, host side.

Mirror of the reference's tools/tokenizer/ReasoningCodec_film/models/transformer_1d_flow.py `Transformer1DModel`
(:162-386) in the configuration models/model_config.json pins (norm_type 'ada_norm_single', 'gelu-approximate',
attention_bias, no cross attention) over models/attention.py `BasicTransformerBlock` (:97-420).  The reference builds these
from diffusers (Attention, FeedForward, TimestepEmbedding, SinusoidalPositionalEmbedding: not vendored, not installed ->
restated from the call sites, PARITY UNPINNED, see oracle/codec_model_oracle.py); parameter names follow diffusers'
modules so a reference checkpoint's `cfm_wrapper.estimator.*` keys load: `proj_in.ffn_1/ffn_2`, `transformer_blocks.N.
{scale_shift_table, attn1.to_q/to_k/to_v/to_out.0, ff.net.0.proj, ff.net.2}`, `scale_shift_table`, `proj_out.*`,
`adaln_single.{emb.timestep_embedder.linear_1/2, linear}`.

Per layer on the device (6 GEMM launches + 1 attention + 3 tiny vector ops; the reference: ~25 PyTorch ops):
  adaLN vectors (table + t_emb, 1 + scale)              ua2_ew_fma                         attention.py:308-310
  LayerNorm * (1 + scale) + shift -> q|k|v + bias       ua2_linear NORM(LayerNorm) / QKV   :311-319, attn1
  dense softmax attention over the T frames             ua2_attn (rows see all positions)  attn1 (non-causal SDPA)
  to_out + bias, * gate, + residual                     ua2_linear CAST / RESIDUAL(scale)  :345-349
  LayerNorm * (1 + scale) + shift -> ff.net.0 + GELU    ua2_linear NORM / GELU(tanh)       :388-390, ff
  ff.net.2 + bias, * gate, + residual                   ua2_linear CAST / RESIDUAL(scale)  :401-405
"""
import math
import os

import torch
import torch.nn as nn

from ..... import ops
from ....._lib import ATTN_BF16_QP, EPI_GELU, EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, EW_SILU, GELU_TANH, ROPE_NONE
from ._dense import DenseKV, PackedLinear


class ProjectLayer(nn.Module):
    """transformer_1d_flow.py:19-33: Conv1d(k, padding k//2) over time, * k^-0.5, Linear."""

    def __init__(self, hidden_size, filter_size, kernel_size=1):
        super().__init__()
        self.kernel_size = kernel_size
        self.ffn_1 = nn.Conv1d(hidden_size, filter_size, kernel_size, padding=kernel_size // 2)
        self.ffn_2 = nn.Linear(filter_size, filter_size)

    def prepare(self, dtype):
        k = self.kernel_size
        w = self.ffn_1.weight.detach().float()                      # [Cout, Cin, k]
        s = k ** -0.5                                               # folded into the conv taps and bias (:31)
        self._taps = [PackedLinear(w[:, :, j].contiguous(), self.ffn_1.bias if j == 0 else None, dtype, scale=s) for j in range(k)]
        self._lin = PackedLinear(self.ffn_2.weight, self.ffn_2.bias, dtype)

    def run(self, x, B, T):
        """x [B*T, Cin] fp32 rows -> [B*T, Cout].  The k-tap conv is k accumulated GEMMs over row-shifted views of a zero-padded
        copy (tap j reads frames t + j - k//2).  All B sequences go through ONE launch per tap: the padded copies lie back to back
        as [B * (T + 2 pad), Cin] rows, tap j multiplies rows j .. j + B (T + 2 pad) - 2 pad of it, and the 2 pad rows per sequence
        whose window straddles two sequences are computed and dropped (a row's value never depends on its neighbours in a
        GEMM)."""
        k, Cin = self.kernel_size, x.shape[1]
        pad = k // 2
        Tp = T + 2 * pad
        xp = torch.zeros(B, Tp, Cin, dtype=torch.float32, device=x.device)
        xp[:, pad:pad + T] = x.view(B, T, Cin)
        xf = xp.view(B * Tp, Cin)
        Mv = B * Tp - 2 * pad                                          # rows whose k-frame window lies inside the buffer
        yf = torch.empty(B * Tp, self._taps[0].N, dtype=torch.float32, device=x.device)
        yv = yf[:Mv]
        for j, tap in enumerate(self._taps):
            src = xf[j:j + Mv]
            if j == 0:
                tap(src, y=yv, M=Mv)
            else:
                tap(src, epilogue=EPI_RESIDUAL, resid=yv, y=yv, M=Mv)
        y = yf.view(B, Tp, -1)[:, :T].reshape(B * T, -1)               # row (b, t) of the flat result = output frame t of sequence b
        return self._lin(y)


class _Attention(nn.Module):
    def __init__(self, dim, bias=True):
        super().__init__()
        self.to_q, self.to_k, self.to_v = (nn.Linear(dim, dim, bias=bias) for _ in range(3))
        self.to_out = nn.ModuleList([nn.Linear(dim, dim, bias=True), nn.Identity()])


class _GELUProj(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)


class _FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])


class BasicTransformerBlock(nn.Module):
    packed_handoff = True      # tests switch it off to compare against the route through the consumer's own prep launch

    def __init__(self, dim, num_attention_heads, attention_head_dim, attention_bias=True, norm_eps=1e-6):
        super().__init__()
        self.dim, self.heads, self.head_dim, self.eps = dim, num_attention_heads, attention_head_dim, norm_eps
        self.attn1 = _Attention(dim, bias=attention_bias)
        self.ff = _FeedForward(dim)
        self.scale_shift_table = nn.Parameter(torch.randn(6, dim) / dim ** 0.5)

    def prepare(self, dtype):
        a = self.attn1
        cat = lambda ts: torch.cat([t.detach().float() for t in ts], 0)
        bias = cat([a.to_q.bias, a.to_k.bias, a.to_v.bias]) if a.to_q.bias is not None else None
        self._p = dict(qkv=PackedLinear(cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]), bias, dtype),
                       out=PackedLinear(a.to_out[0].weight, a.to_out[0].bias, dtype),
                       ff1=PackedLinear(self.ff.net[0].proj.weight, self.ff.net[0].proj.bias, dtype),
                       ff2=PackedLinear(self.ff.net[2].weight, self.ff.net[2].bias, dtype),
                       table=self.scale_shift_table.detach().float().contiguous().view(-1), dtype=dtype)

    ln_handover = True         # tests / A-B: False keeps the consumer's own LayerNorm prep launches (env UA2_DIT_NO_LN_HANDOVER=1 likewise)

    def run(self, h, ts, kv: DenseKV, mod=None, mod1=None, x_packed=None, nxt=None):
        """h [B*T, D] fp32 rows (in place); ts [6*D] = adaln_single.linear(silu(t_emb)) of this step (one timestep for the
        whole batch, as solve_euler passes it).  mod / mod1: this block's (6, D) slices of `table + ts` and `1 + (table + ts)`
        when the caller computed them for all blocks in two launches (same operations, same order).
        x_packed: this block's q|k|v operand, LayerNorm-ed and packed by the previous block's FF2 launch (LayerNorm hand-over,
        ua2hip.h y_ln_w); nxt = (1 + scale, shift) of the NEXT block's first LayerNorm: this block's FF2 then builds that operand.
        Returns (h, operand for the next block or None)."""
        p, D = self._p, self.dim
        if mod is None:
            mod = ops.ew_fma(p["table"], c=ts)                               # table + timestep  -> (6, D)
            mod1 = ops.ew_fma(mod, beta=1.0)
        sh_a, _, g_a, sh_m, _, g_m = (mod[i * D:(i + 1) * D] for i in range(6))
        w_a, w_m = mod1[D:2 * D], mod1[4 * D:5 * D]                          # 1 + scale
        M = h.shape[0]
        q = torch.empty(M, D, dtype=torch.float32, device=h.device)
        if x_packed is not None:
            p["qkv"](None, M=M, x_packed=x_packed, epilogue=EPI_QKV_ROPE, rope_mode=ROPE_NONE, row_pos=kv.row_pos, row_seq=kv.row_seq, q_out=q, kv=kv.geom)
        else:
            p["qkv"](h, epilogue=EPI_QKV_ROPE, norm=(w_a, sh_a, self.eps), rope_mode=ROPE_NONE, row_pos=kv.row_pos, row_seq=kv.row_seq,
                     q_out=q, kv=kv.geom)
        # scratch for the K split of FF2 (ua2hip.h split_ws: K = 4 D on a 1000-row problem leaves 3/4 of the device's workgroup slots
        # empty; four slabs side by side + a combine launch, fixed order).  One buffer per K/V plan, shared by the blocks.
        if getattr(kv, "split_ws", None) is None or kv.split_ws.numel() < 4 * M * D:
            kv.split_ws = torch.empty(4 * M * D, dtype=torch.float32, device=h.device)
        if self.packed_handoff and kv.groups is not None and M > 16 and D % 32 == 0:
            # bf16 plan at many rows: attention and GELU write their consumer's operand in fragment order (same rounding as
            # the consumer's own prep launch would apply: identical bits, two launches less per layer)
            ws_o, ws_f = ops.linear_workspace(p["dtype"], M, D, h.device), ops.linear_workspace(p["dtype"], M, p["ff2"].K, h.device)
            kv.attend(q, y_packed=ws_o)
            # LayerNorm hand-over (order-free plan; round 6): the o-projection and FF2 run as K slabs whose combine ALSO normalises,
            # modulates and packs the rows for FF1 / the next block's q|k|v — the two LayerNorm prep launches per block go.  Asked
            # of the library once per (plan, launch shape): ua2_linear_order_free_accepts.
            want = bool(self.ln_handover and p["out"].sum_order != 0 and not os.environ.get("UA2_DIT_NO_LN_HANDOVER"))
            cached = getattr(kv, "ln_ok", None)
            if cached is None or cached[0] != want:
                probe = dict(M=M, epilogue=EPI_RESIDUAL, resid=h, y=h, split_ws=kv.split_ws, y_ln=(w_m, sh_m, self.eps), probe=True)
                cached = kv.ln_ok = (want, bool(want and p["out"](None, x_packed=ws_o, out_scale=g_a, y_packed=ws_o, **probe) and
                                               p["ff2"](None, x_packed=ws_f, out_scale=g_m, y_packed=ws_o, **probe)))
            ok = cached[1]
            nxt_packed = None
            if ok:
                ws_n = ops.linear_workspace(p["dtype"], M, D, h.device)
                p["out"](None, M=M, x_packed=ws_o, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_a, y=h, split_ws=kv.split_ws,
                         y_ln=(w_m, sh_m, self.eps), y_packed=ws_n)
                p["ff1"](None, M=M, x_packed=ws_n, epilogue=EPI_GELU, act_kind=GELU_TANH, y_packed=ws_f)
                if nxt is not None:
                    nxt_packed = ops.linear_workspace(p["dtype"], M, D, h.device)
                    p["ff2"](None, M=M, x_packed=ws_f, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_m, y=h, split_ws=kv.split_ws,
                             y_ln=(nxt[0], nxt[1], self.eps), y_packed=nxt_packed)
                else:
                    p["ff2"](None, M=M, x_packed=ws_f, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_m, y=h, split_ws=kv.split_ws)
                return h, nxt_packed
            p["out"](None, M=M, x_packed=ws_o, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_a, y=h)
            p["ff1"](h, epilogue=EPI_GELU, norm=(w_m, sh_m, self.eps), act_kind=GELU_TANH, y_packed=ws_f)
            p["ff2"](None, M=M, x_packed=ws_f, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_m, y=h, split_ws=kv.split_ws)
            return h, None
        o = kv.attend(q)
        p["out"](o, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_a, y=h)
        f = p["ff1"](h, epilogue=EPI_GELU, norm=(w_m, sh_m, self.eps), act_kind=GELU_TANH)
        p["ff2"](f, epilogue=EPI_RESIDUAL, resid=h, out_scale=g_m, y=h, split_ws=kv.split_ws)
        return h, None


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, dim):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(in_channels, dim), nn.Linear(dim, dim)


class _CombinedFlowEmbeddings(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.flow_t_size = 512
        self.timestep_embedder = _TimestepEmbedding(self.flow_t_size, dim)


class AdaLayerNormSingleFlow(nn.Module):
    """transformer_1d_flow.py:87-115."""

    def __init__(self, dim):
        super().__init__()
        self.emb = _CombinedFlowEmbeddings(dim)
        self.linear = nn.Linear(dim, 6 * dim, bias=True)

    def prepare(self, dtype):
        te = self.emb.timestep_embedder
        self._p = dict(l1=PackedLinear(te.linear_1.weight, te.linear_1.bias, dtype), l2=PackedLinear(te.linear_2.weight, te.linear_2.bias, dtype),
                       lin=PackedLinear(self.linear.weight, self.linear.bias, dtype))

    def sinusoid(self, t: float):
        """:57-72: the 512-entry cos | sin vector of the step's time t in [0, 1] (host side: the Euler schedule is host-side)."""
        half = self.emb.flow_t_size // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        args = torch.tensor([t], dtype=torch.float32)[:, None] * freqs[None] * 1000
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)

    def run(self, emb):
        """emb [1, 512] device (sinusoid(t)) -> (ts [6D], embedded [D])."""
        p = self._p
        e = p["l2"](ops.ew_act(p["l1"](emb), EW_SILU))
        return p["lin"](ops.ew_act(e, EW_SILU)).view(-1), e.view(-1)


# The hyper-parameters the released codec pins in models/model_config.json (what `from_config` needs of them).
RELEASED_CONFIG = dict(num_attention_heads=24, attention_head_dim=64, in_channels=1040, out_channels=136, num_layers=32,
                       attention_bias=True, norm_eps=1e-6, norm_type="ada_norm_single", activation_fn="gelu-approximate",
                       cross_attention_dim=None)


class Transformer1DModel(nn.Module):
    # Summation-order contract of the DiT's GEMMs (include/ua2hip.h, ua2_linear_args.sum_order).  Nothing downstream relies on a row
    # of the velocity field having the same bits at every row count — the reference's own SDPA / cuBLAS calls do not have that
    # property either — so the bf16 plan takes the order-free 256-row-tile kernel (csrc/ua2_gemm2.hip).  0 (or env UA2_DIT_SUM_ORDER=0)
    # restores the row-invariant kernels: with them a window's latent has the same bits alone and inside a batch of windows
    # (tests/test_gpu_codec_model.py uses that to prove the batching logic exactly).
    sum_order = 1

    def __init__(self, num_attention_heads=24, attention_head_dim=64, in_channels=1040, out_channels=136, num_layers=32,
                 attention_bias=True, norm_eps=1e-6, num_positional_embeddings=3000, **unused_config):
        super().__init__()
        D = num_attention_heads * attention_head_dim
        self.inner_dim, self.heads, self.head_dim = D, num_attention_heads, attention_head_dim
        self.in_channels, self.out_channels, self.max_pos = in_channels, out_channels, num_positional_embeddings
        self.proj_in = ProjectLayer(in_channels, D, kernel_size=3)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(D, num_attention_heads, attention_head_dim, attention_bias, norm_eps)
                                                 for _ in range(num_layers)])
        self.scale_shift_table = nn.Parameter(torch.randn(2, D) / D ** 0.5)
        self.proj_out = ProjectLayer(D, out_channels, kernel_size=3)
        self.adaln_single = AdaLayerNormSingleFlow(D)
        self._ready = False

    @classmethod
    def from_config(cls, path_or_dict):
        import json
        cfg = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
        if (cfg.get("norm_type", "ada_norm_single") != "ada_norm_single" or cfg.get("activation_fn", "gelu-approximate") != "gelu-approximate"
                or cfg.get("cross_attention_dim") is not None):
            raise NotImplementedError("only the released codec's DiT configuration (models/model_config.json) is built")
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    def prepare(self, dtype=torch.bfloat16):
        dev = self.scale_shift_table.device
        if dev.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback); move the model to cuda")
        self.proj_in.prepare(dtype); self.proj_out.prepare(dtype); self.adaln_single.prepare(dtype)
        for b in self.transformer_blocks:
            b.prepare(dtype)
        order = int(os.environ.get("UA2_DIT_SUM_ORDER", self.sum_order)) if dtype == torch.bfloat16 else 0
        self._order = order
        for b in self.transformer_blocks:
            for k in ("qkv", "out", "ff1", "ff2"):
                b._p[k].sum_order = order
        for lin in (*self.proj_in._taps, self.proj_in._lin, *self.proj_out._taps, self.proj_out._lin):
            lin.sum_order = order
        self._table_all = torch.cat([b._p["table"] for b in self.transformer_blocks]).contiguous()
        D = self.inner_dim
        pos = torch.arange(self.max_pos).unsqueeze(1).float()                 # diffusers SinusoidalPositionalEmbedding (:232)
        div = torch.exp(torch.arange(0, D, 2).float() * (-math.log(10000.0) / D))
        pe = torch.zeros(self.max_pos, D)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        self._pe = pe.to(dev).contiguous()
        self._table = self.scale_shift_table.detach().float().contiguous().view(-1)
        self._dtype, self._ready, self._kv, self._kvs, self._graphs, self._emb_cache = dtype, True, None, {}, {}, {}
        # recordings made elsewhere over this plan (BASECFM.solve_euler's whole-solve graphs) hold raw pointers to the packed weights,
        # K/V plans and step embeddings replaced above: they compare this counter and drop themselves
        self._plan_gen = getattr(self, "_plan_gen", 0) + 1
        return self

    def _forward_impl(self, hidden_states, emb):
        B, T, Cin = hidden_states.shape
        D = self.inner_dim
        x = hidden_states.reshape(B * T, Cin).float().contiguous()
        h = self.proj_in.run(x, B, T)
        h = ops.ew_fma(h, c=self._pe[:T])                                     # + pos_embed (:338); the modulo broadcast repeats it per batch element
        ts, e = self.adaln_single.run(emb)
        mod_all = ops.ew_fma(self._table_all, c=ts)                           # every block's table + timestep in one launch (was 3 per block)
        mod1_all = ops.ew_fma(mod_all, beta=1.0)
        xp = None
        nb = len(self.transformer_blocks)
        for l, blk in enumerate(self.transformer_blocks):
            nxt = None
            if l + 1 < nb:                                                    # (1 + scale, shift) of the next block's first LayerNorm
                o = (l + 1) * 6 * D
                nxt = (mod1_all[o + D:o + 2 * D], mod_all[o:o + D])
            _, xp = blk.run(h, ts, self._kv, mod_all[l * 6 * D:(l + 1) * 6 * D], mod1_all[l * 6 * D:(l + 1) * 6 * D], x_packed=xp, nxt=nxt)
        mod = ops.ew_fma(self._table, c=e)                                    # (2, D): scale_shift_table + embedded_timestep  :378
        shift, scale = mod[:D], mod[D:]
        hn = ops.layernorm_rows(h, None, None, 1e-6)                          # norm_out :379
        hm = ops.ew_fma(hn, b=ops.ew_fma(scale, beta=1.0), c=shift)           # * (1 + scale) + shift  :381
        return self.proj_out.run(hm, B, T).view(B, T, self.out_channels)

    def step_embedding(self, timestep, dev):
        """sinusoid(t) on the device, cached per t (the Euler schedule revisits the same few times window after window)."""
        key = float(timestep)
        e = self._emb_cache.get(key)
        if e is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("Transformer1DModel: time embedding of t=%r requested for the first time inside a graph capture "
                                   "(warm the step up outside the capture first)" % key)
            e = self._emb_cache[key] = self.adaln_single.sinusoid(key).to(dev)
        return e

    def ensure_plan(self, B, T, dev):
        """K/V pools and row tables for B sequences of T frames (allocations and one host sync: outside any capture).  One plan per
        (B, T), kept for the model's lifetime: recorded graphs — this module's per-step ones and the whole-solve ones of
        BASECFM.solve_euler — hold raw pointers into the plan they were captured with (a handful of shapes per deployment:
        2 x codec_batch sequences of 500 frames)."""
        kv = self._kvs.get((B, T))
        if kv is None:
            kv = self._kvs[(B, T)] = DenseKV(B, T, self.heads, self.head_dim, self._dtype, dev)
            # order-free plan: attention as the reference's bf16-autocast SDPA computes it (q and softmax weights in bf16)
            kv.attn_flags = ATTN_BF16_QP if (self._order and os.environ.get("UA2_DIT_ATTN_SPLIT") is None) else 0
        self._kv = kv

    @torch.inference_mode()
    def forward(self, hidden_states, timestep: float, use_graph: bool = True):
        """hidden_states (B, T, in_channels) fp32, one timestep for the whole batch -> (B, T, out_channels) fp32
        (transformer_1d_flow.py:279-386 `.sample`).  The ~400 launches of a step are captured once per (B, T) into a HIP
        graph and replayed (the Euler loop calls this 10-50 times per window with identical shapes; issued one by one from
        Python the step was host-bound: 39 ms at the released size against ~8 ms of kernel time)."""
        if not self._ready:
            self.prepare()
        B, T, Cin = hidden_states.shape
        dev = hidden_states.device
        if torch.cuda.is_current_stream_capturing():
            if (B, T) not in self._kvs:
                raise RuntimeError("Transformer1DModel: no K/V plan for this shape inside a graph capture (run the step once outside it)")
            self._kv = self._kvs[(B, T)]
        else:
            self.ensure_plan(B, T, dev)
        if torch.cuda.is_current_stream_capturing():
            # inside somebody else's capture (BASECFM.solve_euler records a whole window): the launches of this step become nodes
            # of THAT graph; the step's time embedding must already be on the device (no host copy may be recorded)
            return self._forward_impl(hidden_states, self.step_embedding(timestep, dev))
        emb = self.adaln_single.sinusoid(float(timestep))
        if not use_graph:
            return self._forward_impl(hidden_states, emb.to(dev))
        g = self._graphs.get((B, T))
        if g is None:
            x_in = torch.empty(B, T, Cin, dtype=torch.float32, device=dev)
            e_in = torch.empty(1, emb.shape[1], dtype=torch.float32, device=dev)
            x_in.copy_(hidden_states); e_in.copy_(emb)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                     # warm-up outside capture: one-time kernel attributes, allocator pools
                self._forward_impl(x_in, e_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y = self._forward_impl(x_in, e_in)
            g = self._graphs[(B, T)] = (graph, x_in, e_in, y)
        graph, x_in, e_in, y = g
        x_in.copy_(hidden_states)
        e_in.copy_(emb, non_blocking=True)
        graph.replay()
        return y.clone()
