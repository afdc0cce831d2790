"""Host-side helpers shared by the codec's neural stages (AudioThinking encoder, flow-matching DiT): a packed Linear
(`nn.Linear` call sites -> one ua2_linear launch with the surrounding norm / bias / activation / gated residual fused),
a dense (non-causal) attention over the paged K/V layout, and torch's own nearest-neighbour index rule.  No torch math on
the data path; torch owns memory and builds index lists."""
import torch
import torch.nn.functional as F

from ..... import ops
from ....._lib import (EPI_GELU, EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, NORM_LAYERNORM, PRO_CAST, PRO_NORM, ROPE_NONE,
                       UA2_PAGE)


def folded_weight(lin):
    """Effective fp32 weight of an nn.Linear, weight-norm (parametrizations) folded: w = g * v / ||v||_row."""
    if hasattr(lin, "parametrizations") and "weight" in lin.parametrizations:
        p = lin.parametrizations.weight
        return torch._weight_norm(p.original1.detach().float(), p.original0.detach().float(), 0)
    return lin.weight.detach().float()


class PackedLinear:
    """weight [N, K] (+ bias [N]) packed once for `dtype`; call = one ua2_linear launch."""

    def __init__(self, weight, bias, dtype, scale=None):
        w = weight.detach().float()
        b = bias.detach().float() if bias is not None else None
        if scale is not None:
            w, b = w * scale, (b * scale if b is not None else None)
        self.N, self.K = w.shape
        self.dtype = dtype
        self.w = ops.pack_linear(w.contiguous(), dtype)
        self.b = b.contiguous() if b is not None else None
        self.sum_order = 0          # ua2hip.h sum_order: callers outside every row-invariance contract (the DiT) set UA2_SUM_ORDER_FREE

    def __call__(self, x, *, epilogue=EPI_STORE, norm=None, resid=None, out_scale=None, act_kind=0, w1=None, y=None, M=None, **kw):
        """x [M, K] fp32 contiguous rows.  norm = (w, b, eps) -> LayerNorm prologue (F.layer_norm then * w + b)."""
        M = M or x.shape[0]
        dev = self.w.device
        packed_in, packed_out = kw.get("x_packed") is not None, kw.get("y_packed") is not None   # operand handed over / handed on in fragment order
        if y is None and epilogue != EPI_QKV_ROPE and not packed_out:  # QKV: the results go to q_out and the K/V pools
            y = torch.empty(M, self.N, dtype=torch.float32, device=dev)
        ws = ops.linear_workspace(self.dtype, M, self.K, dev) if M > 16 and not packed_in else None
        extra = {}
        if norm is not None:
            extra = dict(prologue=PRO_NORM, norm_w=norm[0], norm_b=norm[1], eps=norm[2], norm_kind=NORM_LAYERNORM)
        if w1 is not None:
            extra.update(w1=w1.w, bias1=w1.b)
        kw.setdefault("sum_order", self.sum_order)
        probe = kw.pop("probe", False)     # True: launch nothing, return whether the order-free kernel would take this launch (ua2hip.h)
        a = ops.linear(dtype=self.dtype, M=M, N=self.N, K=self.K, w0=self.w, epilogue=epilogue, x=x, ldx=self.K, y=y, ldy=(self.N if y is not None else 0), resid=resid,
                       ldr=(self.N if resid is not None else None), out_scale=out_scale, bias=self.b, act_kind=act_kind, workspace=ws, launch=not probe,
                       **extra, **kw)
        if probe:
            import ctypes
            from ....._lib import lib
            return bool(lib.ua2_linear_order_free_accepts(ctypes.byref(a)))
        return y if y is not None else kw.get("q_out")


class DenseKV:
    """Paged K/V pools for B sequences of up to T positions, one layer at a time (re-used by every layer: a layer's K/V
    are dead once its attention has run)."""

    def __init__(self, B, T, n_head, head_size, dtype, device):
        self.max_pages = (T + UA2_PAGE - 1) // UA2_PAGE
        shape = (B * self.max_pages, n_head, UA2_PAGE, head_size)
        self.k = torch.zeros(shape, dtype=dtype, device=device)
        self.v = torch.zeros(shape, dtype=dtype, device=device)
        self.table = torch.arange(B * self.max_pages, dtype=torch.int32, device=device).view(B, self.max_pages)
        self.geom = ops.kv_geom(self.k, self.v, self.table, n_head, n_head, head_size)
        i32 = dict(dtype=torch.int32, device=device)
        self.row_pos = torch.arange(T, **i32).repeat(B)                       # row r = (b, t): position t ...
        self.row_seq = torch.arange(B, **i32).repeat_interleave(T)            # ... of sequence b
        self.all_pos = torch.full((B * T,), T - 1, **i32)                     # non-causal: every row sees positions 0..T-1
        self.dtype, self.B, self.T = dtype, B, T
        self.attn_flags = 0            # ua2hip.h UA2_ATTN_BF16_QP: set by callers whose reference runs SDPA under bf16 autocast (the DiT)
        # bf16: the MFMA flash form of ua2_attn (K/V pages staged once per 64 query rows instead of once per row)
        # (128 query rows per workgroup at head size 64: the DiT step 6.76 -> 6.61 ms against 64)
        self.groups = ops.attn_groups(self.all_pos.cpu().numpy(), self.row_seq.cpu().numpy(), n_head, n_head, device,
                                      q_tiles=8 if head_size == 64 else None) if dtype == torch.bfloat16 else None

    def attend(self, q, y_packed=None):
        """q [B*T, n_head*hs] fp32 -> softmax(q K^T / sqrt(hs)) V, every row over all T positions of its sequence; with
        y_packed (a ua2_linear workspace) the rows are written in the consumer's operand order instead and nothing is returned."""
        y = torch.empty_like(q) if y_packed is None else None
        ops.attn(dtype=self.dtype, R=q.shape[0], q=q, row_pos=self.all_pos, row_seq=self.row_seq, kv=self.geom, y=y, groups=self.groups,
                 y_packed=y_packed, flags=self.attn_flags)
        return y


def nearest_indices(T_in, scale_factor, device):
    """Source index of every output step of F.interpolate(mode='nearest', scale_factor=s) along time — obtained from
    torch's own rule by interpolating an index ramp (AudioDiffusion1D.py:450,512,590)."""
    ramp = torch.arange(T_in, dtype=torch.float32).view(1, 1, T_in)
    return F.interpolate(ramp, scale_factor=scale_factor, mode="nearest").view(-1).to(torch.int32).to(device)
