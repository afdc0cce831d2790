"""Op-level Python wrappers over the C ABI (torch is only the owner of device memory and the
stream).  Each wrapper names the reference call site it stands in for; the arithmetic lives in
csrc/*.hip.  Used by the parity tests and by the module mirrors in llm_models/.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import (EPI_QKV_ROPE, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST, PRO_NORM, UA2_BF16,
                   UA2_F32, UA2_PAGE, AttnArgs, KvGeom, LinearArgs, check, lib)

_CODES = {torch.float32: UA2_F32, torch.bfloat16: UA2_BF16}


def dtype_code(dt):
    try:
        return _CODES[dt]
    except KeyError:
        raise ValueError(f"uniaudio2_amd supports torch.float32 and torch.bfloat16 weights, got {dt}")


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor expected"
    return t.data_ptr()


def packed_elems(dtype, N, K):
    return lib.ua2_packed_elems(dtype_code(dtype), N, K)


def pack_linear(weight, dtype, transposed=False, rope_head_size=0):
    """nn.Linear.weight [N,K] (or [K,N] with transposed=True) -> MFMA-fragment-ordered buffer of `dtype`."""
    assert weight.dim() == 2 and weight.is_cuda
    w = weight.contiguous()
    if w.dtype not in _CODES:
        w = w.float()
    N, K = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    out = torch.empty(packed_elems(dtype, N, K), dtype=dtype, device=w.device)
    check(lib.ua2_pack_linear(ptr(w), dtype_code(w.dtype), int(transposed), N, K, ptr(out), dtype_code(dtype),
                              rope_head_size, stream()), "ua2_pack_linear")
    return out


def kv_geom(k_pool, v_pool, page_table, n_head, n_kv, head_size, ring_pages=0):
    g = KvGeom()
    g.ring_pages = ring_pages
    g.k_pool, g.v_pool, g.page_table = ptr(k_pool), ptr(v_pool), ptr(page_table)
    g.max_pages = page_table.shape[-1] if page_table is not None else 0
    g.n_kv, g.n_head, g.head_size = n_kv, n_head, head_size
    return g


def linear(*, dtype, M, N, K, w0, prologue=PRO_CAST, epilogue=EPI_STORE, x=None, ldx=None, norm_w=None, eps=1e-5,
           w1=None, y=None, ldy=None, resid=None, ldr=None, part_max=None, part_idx=None,
           forbid=None, row_pos=None, row_seq=None, rope_cos=None, rope_sin=None, q_out=None, kv=None, launch=True,
           norm_b=None, norm_kind=0, out_scale=None, rope_mode=0, workspace=None, bias=None, bias1=None, act_kind=0,
           y_packed=None, x_packed=None, y_norm_w=None, y_h=None, ldh=0, y_ssq=None, x_h=None, x_ssq=None, split_ws=None, sum_order=0, y_ln=None, range_ws=None):
    a = LinearArgs()
    a.dtype, a.prologue, a.epilogue = dtype_code(dtype), prologue, epilogue
    a.M, a.N, a.K = M, N, K
    a.x, a.ldx = ptr(x), (ldx if ldx is not None else (x.shape[-1] if x is not None else 0))
    a.norm_w, a.eps = ptr(norm_w), eps
    a.w0, a.w1 = ptr(w0), ptr(w1)
    a.y, a.ldy = ptr(y), (ldy if ldy is not None else (y.shape[-1] if y is not None else 0))
    a.resid, a.ldr = ptr(resid), (ldr if ldr is not None else (resid.shape[-1] if resid is not None else 0))
    a.part_max, a.part_idx, a.forbid = ptr(part_max), ptr(part_idx), ptr(forbid)
    a.row_pos, a.row_seq = ptr(row_pos), ptr(row_seq)
    a.rope_cos, a.rope_sin, a.q_out = ptr(rope_cos), ptr(rope_sin), ptr(q_out)
    a.norm_b, a.norm_kind, a.out_scale, a.rope_mode = ptr(norm_b), norm_kind, ptr(out_scale), rope_mode
    a.bias, a.bias1, a.act_kind = ptr(bias), ptr(bias1), act_kind
    a.y_packed, a.x_packed = ptr(y_packed), ptr(x_packed)
    a.y_norm_w, a.y_h, a.y_ssq, a.x_h, a.x_ssq = ptr(y_norm_w), ptr(y_h), ptr(y_ssq), ptr(x_h), ptr(x_ssq)
    a.ldh = ldh or (y_h.shape[-1] if y_h is not None else (x_h.shape[-1] if x_h is not None else 0))
    if workspace is not None:
        a.workspace, a.workspace_bytes = ptr(workspace), workspace.numel() * workspace.element_size()
    if split_ws is not None:
        a.split_ws, a.split_ws_bytes = ptr(split_ws), split_ws.numel() * split_ws.element_size()
    if range_ws is not None:       # scratch of the range split of row-invariant 33-64-row launches (ua2hip.h [v10]): same bits with and without
        a.range_ws, a.range_ws_bytes = ptr(range_ws), range_ws.numel() * range_ws.element_size()
    a.sum_order = sum_order
    if y_ln is not None:                 # (w, b, eps): LayerNorm hand-over of a RESIDUAL launch (ua2hip.h y_ln_w)
        a.y_ln_w, a.y_ln_b, a.y_ln_eps = ptr(y_ln[0]), ptr(y_ln[1]), float(y_ln[2])
    if kv is not None:
        a.kv = kv
    if not launch:
        return a
    check(lib.ua2_linear(C.byref(a), stream()), "ua2_linear")


def linear_workspace(dtype, M, K, device):
    """Scratch that lets ua2_linear take its large-M kernel for up to M rows of width K."""
    n = lib.ua2_linear_workspace_bytes(dtype_code(dtype), M, K)
    return torch.empty(n, dtype=torch.uint8, device=device)


def linear_chain_timed(args_list, iters):
    """Average milliseconds per launch of the given ua2_linear launches, back to back, HIP-event timed."""
    arr = (LinearArgs * len(args_list))(*args_list)
    ms = C.c_float(0.0)
    check(lib.ua2_linear_chain_timed(arr, len(args_list), iters, stream(), C.byref(ms)), "ua2_linear_chain_timed")
    return ms.value / (len(args_list) * iters)


def attn_groups(pos, seq, n_head, n_kv, device, q_tiles=None):
    """Row groups for the MFMA flash form of ua2_attn: pos, seq = host int sequences (position and page-table row of every
    query row of the launch).  Rows of one sequence, ordered by position, are cut into groups of q_tiles * 16
    (q_tiles = 2 with grouped-query heads, 4 otherwise; a caller whose head size has the wider instantiation may ask for 8:
    twice the query rows per staged K / V page).  -> (rows [n, q_tiles*16] int32, seq [n], nkeys [n], q_tiles), device."""
    import numpy as np
    pos, seq = np.asarray(pos, dtype=np.int64), np.asarray(seq, dtype=np.int64)
    qt = q_tiles or (2 if n_head > n_kv else 4)
    if n_head == n_kv and os.environ.get("UA2_ATTN_QTILES"):   # experiment hook (a row's bits do not depend on the grouping)
        qt = int(os.environ["UA2_ATTN_QTILES"])
    per = qt * 16
    order = np.lexsort((pos, seq))
    rows, gseq, nkeys = [], [], []
    start = 0
    while start < len(order):
        s0 = seq[order[start]]
        end = start
        while end < len(order) and seq[order[end]] == s0:
            end += 1
        for c in range(start, end, per):
            idx = order[c:min(c + per, end)]
            rows.append(np.concatenate([idx, np.full(per - len(idx), -1)]))
            gseq.append(s0)
            nkeys.append(int(pos[idx].max()) + 1)
        start = end
    to = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).to(device)
    return to(np.stack(rows)).contiguous(), to(gseq), to(nkeys), qt


def attn(*, dtype, R, q, row_pos, row_seq, kv, y=None, window=0, groups=None, y_packed=None, flags=0):
    """y [R, n_head*hs] fp32 and / or y_packed: the same rows already in ua2_linear's operand order (its consumer then takes
    them as x_packed and needs no prep launch)."""
    a = AttnArgs()
    a.flags = flags
    a.y_packed = ptr(y_packed)
    if groups is not None:
        a.group_rows, a.group_seq, a.group_nkeys = ptr(groups[0]), ptr(groups[1]), ptr(groups[2])
        a.n_groups, a.group_q_tiles = groups[0].shape[0], groups[3]
    a.y = ptr(y)
    a.window = window
    a.dtype, a.R = dtype_code(dtype), R
    a.q, a.row_pos, a.row_seq = ptr(q), ptr(row_pos), ptr(row_seq)
    a.kv = kv
    check(lib.ua2_attn(C.byref(a), stream()), "ua2_attn")


def attn_local(*, dtype, R, q, row_pos, row_seq, kv, y):
    """Short-context attention of the depth decoder (row_pos < 8, first cache page): y [R, n_head*head_size] fp32."""
    a = AttnArgs()
    a.dtype, a.R = dtype_code(dtype), R
    a.q, a.row_pos, a.row_seq, a.y, a.kv = ptr(q), ptr(row_pos), ptr(row_seq), ptr(y), kv
    check(lib.ua2_attn_local(C.byref(a), stream()), "ua2_attn_local")


def embed_frame(dtype, tokens, mask, audio_emb, wte, va):
    M, w = tokens.shape
    Cc = audio_emb.shape[1]
    a = torch.empty(M, Cc, dtype=torch.float32, device=tokens.device)
    t = torch.empty_like(a)
    check(lib.ua2_embed_frame(dtype_code(dtype), M, Cc, w - 1, va, ptr(tokens), ptr(mask), ptr(audio_emb), ptr(wte),
                              ptr(a), ptr(t), None, stream()), "ua2_embed_frame")
    return a, t


def rmsnorm_blend(x, w, eps, other=None, mask=None, col_a=-1, col_b=-1, want_n=False):
    M, Cc = x.shape
    o1 = torch.empty_like(x)
    o2 = torch.empty_like(x) if want_n else None
    check(lib.ua2_rmsnorm_blend(M, Cc, ptr(x), ptr(w), eps, ptr(other), ptr(mask),
                                mask.shape[1] if mask is not None else 0, col_a, col_b, ptr(o1), ptr(o2), None, stream()),
          "ua2_rmsnorm_blend")
    return (o1, o2) if want_n else o1


def argmax_embed(dtype, part_max, part_idx, out_tokens, out_col, emb=None, emb_row_offset=0, next_h=None):
    M, n_part = part_max.shape
    check(lib.ua2_argmax_embed(dtype_code(dtype), M, n_part, ptr(part_max), ptr(part_idx), ptr(out_tokens),
                               out_tokens.shape[1], out_col, ptr(emb), emb_row_offset,
                               emb.shape[1] if emb is not None else 0, ptr(next_h), stream()), "ua2_argmax_embed")


def rvq_encode(x, emb, embT=None):
    """x [N,D] fp32, emb [L,C,D] fp32 -> codes [N,L] int32, quantized [N,D] fp32 (core_vq.py:365-376)."""
    N, D = x.shape
    L, Cc, _ = emb.shape
    if embT is None:
        embT = emb.transpose(1, 2).contiguous()
    codes = torch.empty(N, L, dtype=torch.int32, device=x.device)
    q = torch.empty(N, D, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.ua2_rvq_workspace_bytes(N, L), dtype=torch.uint8, device=x.device)
    check(lib.ua2_rvq_encode(ptr(x), ptr(emb), ptr(embT), N, L, Cc, D, ptr(codes), ptr(q), ptr(ws), ws.numel(), stream()), "ua2_rvq_encode")
    return codes, q


def rvq_decode(codes, emb):
    """codes [N,L] int32, emb [L,C,D] -> [N,D] fp32 (core_vq.py:378-384)."""
    N, L = codes.shape
    _, Cc, D = emb.shape
    out = torch.empty(N, D, dtype=torch.float32, device=emb.device)
    check(lib.ua2_rvq_decode(ptr(codes), ptr(emb), N, L, Cc, D, ptr(out), stream()), "ua2_rvq_decode")
    return out


# ---- codec convolutions ------------------------------------------------------------------------

def pack_conv_weight(w):
    """nn.Conv1d weight [Cout, Cin, K] fp32 -> packed buffer for ua2_conv1d (Cin zero-padded to 16)."""
    Cout, Cin, K = w.shape
    cin_pad = (Cin + 15) // 16 * 16
    wp = torch.zeros(Cout, cin_pad, K, dtype=torch.float32, device=w.device)
    wp[:, :Cin] = w.float()
    return pack_linear(wp.view(Cout, cin_pad * K), torch.float32), K


def convtr_phase_rows(w, stride):
    """nn.ConvTranspose1d weight [Cin, Cout, K] -> [stride * Cout, Cin, M] phase filters of ua2_conv1d's phase mode:
    row phase*Cout + co, taps in descending order (x index q - m  <->  original tap phase + m*stride)."""
    Cin, Cout, K = w.shape
    M = (K + stride - 1) // stride
    wp = torch.zeros(stride, Cout, Cin, M, dtype=torch.float32, device=w.device)
    for ph in range(stride):
        for m in range(M):
            j = ph + m * stride
            if j < K:
                wp[ph, :, :, M - 1 - m] = w[:, :, j].float().t()
    return wp.view(stride * Cout, Cin, M), M


def pack_convtr_weight(w, stride):
    """Packed fp32 phase filters of a transposed conv for the exact ua2_conv1d (see convtr_phase_rows)."""
    rows, M = convtr_phase_rows(w, stride)
    return pack_conv_weight(rows)[0], M


def pack_conv_weight_x3(w_rows):
    """[rows, Cin, K] fp32 filter (rows = Cout, or phases * Cout for a transposed conv) -> (hi, lo) packed bf16 buffers for
    ua2_conv1d precision 1: reduction index (channel group of 32, tap, channel in group), Cin zero-padded to 32."""
    rows, Cin, K = w_rows.shape
    G = (Cin + 31) // 32
    wp = torch.zeros(rows, G * 32, K, dtype=torch.float32, device=w_rows.device)
    wp[:, :Cin] = w_rows.float()
    wk = wp.view(rows, G, 32, K).permute(0, 1, 3, 2).reshape(rows, G * K * 32).contiguous()
    hi = wk.to(torch.bfloat16)
    lo = (wk - hi.float()).contiguous()
    return pack_linear(hi.float().contiguous(), torch.bfloat16), pack_linear(lo, torch.bfloat16)


def conv1d(x, w_packed, K, Cout, *, stride=1, dilation=1, pad_left=0, Tout=None, bias=None, pre_act=0, pre_alpha=None,
           post_act=0, post_alpha=None, residual=None, in_repeat=1, out_phases=1, out_trim_left=0, w_lo=None, fused2=None):
    from ._lib import Conv1dArgs
    B, Cin, Tin = x.shape
    a = Conv1dArgs()
    a.B, a.Cin, a.Cout, a.Tin, a.Tout = B, Cin, Cout, Tin, Tout
    a.K, a.stride, a.dilation, a.pad_left = K, stride, dilation, pad_left
    a.in_repeat, a.out_phases, a.out_trim_left = in_repeat, out_phases, out_trim_left
    a.pre_act, a.post_act = pre_act, post_act
    y = torch.empty(B, Cout, Tout, dtype=torch.float32, device=x.device)
    a.x, a.w, a.bias = ptr(x), ptr(w_packed), ptr(bias)
    a.pre_alpha, a.post_alpha = ptr(pre_alpha), ptr(post_alpha)
    a.post_alpha_n = post_alpha.numel() if post_alpha is not None else 0
    a.residual, a.y = ptr(residual), ptr(y)
    if w_lo is not None:                  # bf16 x 3 form: (w_packed, w_lo) = pack_conv_weight_x3(...)
        a.w_lo, a.precision = ptr(w_lo), 1
    if fused2 is not None:                # (w2_hi, w2_lo, bias2, alpha2): the 1 x 1 conv + PReLU + residual of a residual unit, fused
        a.w2, a.w2_lo, a.bias2, a.alpha2 = ptr(fused2[0]), ptr(fused2[1]), ptr(fused2[2]), ptr(fused2[3])
    check(lib.ua2_conv1d(C.byref(a), stream()), "ua2_conv1d")
    return y


class TC:
    """Activation in the decode side's layout: two bf16 planes hi = RNE(x), lo = RNE(x - hi), each [B, T, C] (time-major,
    channels contiguous) — include/ua2hip.h, ua2_conv1d_tc.  `planes` is one (2, B, T, C) bf16 tensor."""

    def __init__(self, planes):
        assert planes.dim() == 4 and planes.shape[0] == 2 and planes.dtype == torch.bfloat16 and planes.is_contiguous()
        self.planes = planes

    @property
    def hi(self):
        return self.planes[0]

    @property
    def lo(self):
        return self.planes[1]

    @property
    def shape(self):                      # (B, C, T): what the fp32 tensor it stands for would report
        _, B, T, Cc = self.planes.shape
        return (B, Cc, T)

    @staticmethod
    def empty(B, Cc, T, device):
        return TC(torch.empty(2, B, T, Cc, dtype=torch.bfloat16, device=device))


def tc_w2_order(w2):
    """[Cout, C, 1] filter of a fused residual unit's 1 x 1 conv -> the same filter with its input channels in the K order the
    ua2_conv1d_tc kernels reduce in (csrc/ua2_convtc.hip, "tc_w2_order"): inside each group of 32, k' = 8 g + e holds channel
    4 g + e (e < 4) or 16 + 4 g + (e - 4) (e >= 4).  Pack the result with pack_conv_weight_x3 and pass it as `fused2`."""
    Cc = w2.shape[1]
    assert Cc % 32 == 0
    k = torch.arange(32)
    g, e = k // 8, k % 8
    ch = torch.where(e < 4, 4 * g + e, 16 + 4 * g + (e - 4))
    perm = (torch.arange(0, Cc, 32).view(-1, 1) + ch.view(1, -1)).reshape(-1).to(w2.device)
    return w2[:, perm].contiguous()


def tc_pack(x):
    """fp32 [B, C, T] -> TC (ua2_tc_pack)."""
    B, Cc, T = x.shape
    out = TC.empty(B, Cc, T, x.device)
    check(lib.ua2_tc_pack(ptr(x.contiguous()), ptr(out.hi), ptr(out.lo), B, Cc, T, stream()), "ua2_tc_pack")
    return out


def tc_unpack(t):
    """TC -> fp32 [B, C, T], x = hi + lo exactly (ua2_tc_unpack)."""
    B, Cc, T = t.shape
    y = torch.empty(B, Cc, T, dtype=torch.float32, device=t.planes.device)
    check(lib.ua2_tc_unpack(ptr(t.hi), ptr(t.lo), ptr(y), B, Cc, T, stream()), "ua2_tc_unpack")
    return y


def conv1d_tc(x, w_hi, w_lo, K, Cout, *, dilation=1, pad_left=0, Tout=None, bias=None, post_act=0, post_alpha=None, in_repeat=1,
              out_phases=1, out_trim_left=0, fused2=None, residual=None, out_f32=False, variant=0):
    """ua2_conv1d_tc: x a TC; (w_hi, w_lo) = pack_conv_weight_x3(filter rows).  Returns a TC, or fp32 [B, Cout, Tout] with
    out_f32 (the waveform).  fused2 = (w2_hi, w2_lo, bias2, alpha2): the 1 x 1 conv + PReLU + residual of a residual unit in
    the same launch (residual = x), with (w2_hi, w2_lo) = pack_conv_weight_x3(tc_w2_order(W2)); residual = a TC added after the
    activation (un-fused second conv).  variant: 0 automatic, 1 plain, 2 pipelined, 3 big-tile."""
    from ._lib import ConvTcArgs
    B, Cin, Tin = x.shape
    a = ConvTcArgs()
    a.B, a.Cin, a.Cout, a.Tin, a.Tout = B, Cin, Cout, Tin, Tout
    a.K, a.dilation, a.pad_left, a.in_repeat = K, dilation, pad_left, in_repeat
    a.out_phases, a.out_trim_left, a.post_act, a.variant = out_phases, out_trim_left, post_act, variant
    a.x_hi, a.x_lo, a.w, a.w_lo, a.bias = ptr(x.hi), ptr(x.lo), ptr(w_hi), ptr(w_lo), ptr(bias)
    a.post_alpha = ptr(post_alpha)
    a.post_alpha_n = post_alpha.numel() if post_alpha is not None else 0
    if fused2 is not None:
        a.w2, a.w2_lo, a.bias2, a.alpha2 = ptr(fused2[0]), ptr(fused2[1]), ptr(fused2[2]), ptr(fused2[3])
    if residual is not None:
        assert residual.shape == (B, Cout, Tout)
        a.res_hi, a.res_lo = ptr(residual.hi), ptr(residual.lo)
    dev = x.planes.device
    if out_f32:
        y = torch.empty(B, Cout, Tout, dtype=torch.float32, device=dev)
        a.y_f32 = ptr(y)
    else:
        y = TC.empty(B, Cout, Tout, dev)
        a.y_hi, a.y_lo = ptr(y.hi), ptr(y.lo)
    check(lib.ua2_conv1d_tc(C.byref(a), stream()), "ua2_conv1d_tc")
    return y


def dwconv1d(x, w, *, stride=1, dilation=1, pad_left=0, Tout=None, bias=None, transposed=False):
    """Depthwise conv / transposed conv: x [B,C,Tin] fp32, w [C,K] fp32 -> [B,C,Tout] (ua2_dwconv1d)."""
    B, Cc, Tin = x.shape
    K = w.shape[-1]
    y = torch.empty(B, Cc, Tout, dtype=torch.float32, device=x.device)
    check(lib.ua2_dwconv1d(ptr(x.contiguous()), ptr(w.contiguous()), ptr(bias), ptr(y), B, Cc, Tin, Tout, K, stride, dilation,
                           pad_left, int(transposed), stream()), "ua2_dwconv1d")
    return y


def avgpool1d(x, k):
    B, Cc, T = x.shape
    y = torch.empty(B, Cc, T // k, dtype=torch.float32, device=x.device)
    check(lib.ua2_avgpool1d(ptr(x), ptr(y), B * Cc, T, k, stream()), "ua2_avgpool1d")
    return y


# ---- codec neural stages: glue ops -----------------------------------------------------------------

def ew_fma(a, b=None, c=None, alpha=1.0, beta=0.0, out=None, n=None):
    """out[i] = alpha * a[i % na] * b[i % nb] + c[i % nc] + beta over n = max numel (ua2_ew_fma)."""
    ts = [t for t in (a, b, c) if t is not None]
    n = n or max(t.numel() for t in ts)
    if out is None:
        big = max(ts, key=lambda t: t.numel())
        out = torch.empty(big.shape, dtype=torch.float32, device=a.device)
    nz = lambda t: t.numel() if t is not None else 0
    check(lib.ua2_ew_fma(ptr(out), n, ptr(a), a.numel(), ptr(b), nz(b), ptr(c), nz(c), float(alpha), float(beta), stream()), "ua2_ew_fma")
    return out


def ew_act(x, act):
    out = torch.empty_like(x)
    check(lib.ua2_ew_act(ptr(out), ptr(x), x.numel(), act, stream()), "ua2_ew_act")
    return out


def gather_rows(x, idx):
    """x [N, C] fp32, idx [R] int32 (negative = zero row) -> [R, C]."""
    R, Cc = idx.numel(), x.shape[-1]
    out = torch.empty(R, Cc, dtype=torch.float32, device=x.device)
    check(lib.ua2_gather_rows(ptr(out), ptr(x), ptr(idx), R, Cc, stream()), "ua2_gather_rows")
    return out


def time_film(params, x, batch_mask, rows_per_batch, gamma_scale):
    R, Cc = x.shape
    out = torch.empty_like(x)
    check(lib.ua2_time_film(ptr(out), ptr(params), ptr(x), ptr(batch_mask), R, rows_per_batch, Cc, float(gamma_scale), stream()),
          "ua2_time_film")
    return out


def layernorm_rows(x, w=None, b=None, eps=1e-5):
    R, Cc = x.shape
    out = torch.empty_like(x)
    check(lib.ua2_layernorm_rows(ptr(out), ptr(x), ptr(w), ptr(b), R, Cc, float(eps), stream()), "ua2_layernorm_rows")
    return out


def qknorm_rope_kv(dtype, qkv, row_pos, row_seq, kv, q_out, qw=None, qb=None, kw=None, kb=None, eps=1e-5, cos=None, sin=None,
                   rot_dim=0):
    check(lib.ua2_qknorm_rope_kv(dtype_code(dtype), ptr(qkv), qkv.shape[0], ptr(row_pos), ptr(row_seq), ptr(qw), ptr(qb), ptr(kw),
                                 ptr(kb), float(eps), ptr(cos), ptr(sin), rot_dim, ptr(q_out), C.byref(kv), stream()),
          "ua2_qknorm_rope_kv")
