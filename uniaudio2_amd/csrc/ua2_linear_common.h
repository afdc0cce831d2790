// Pieces shared by the two weight-streaming GEMM kernels (ua2_linear.hip: general M;
// ua2_gemv.hip: the decode regime).  Internal, not part of the C ABI.
#pragma once
#include "ua2_common.h"

// ---- launch geometry of the decode-regime kernel (ua2_gemv.hip) --------------------------------
// A function of (dtype, N, K, matrices per launch) only.  It fixes the order in which a row's dot
// product is summed: `waves` partial sums over contiguous chunk ranges [w*nchunks/waves,
// (w+1)*nchunks/waves), each an MFMA chain starting from zero, added in wave order.  The large-M kernel
// (ua2_gemm.hip) reproduces exactly that order, which is what makes the two bit-identical per row.
struct ua2_gemv_geometry { int waves, cpw; };
ua2_gemv_geometry ua2_pick_gemv_geometry(int dtype, int N, int K, int nt);
int ua2_gemv_rows_per_tile(int dtype, int K);
// large-M path; returns 1 when not applicable (no workspace, ATTN prologue, few rows)
// force: 0 = only when M spans more than one row tile; 3 = whenever possible; 4 / 5 = likewise, skinny / tiled form
int ua2_gemm_try_launch(const ua2_linear_args& a, hipStream_t s, int force);
// batched-decode form (ua2_skinny.hip): 0 = launched, 1 = shape outside its table (the older skinny kernel serves it)
int ua2_skinny2_try_launch(const ua2_linear_args& a, ua2_gemv_geometry geo, hipStream_t s);
// order-free form (ua2_gemm2.hip: 256-row tiles, one chain over K): 0 = launched, 1 = launch outside its forms; operand already packed
int ua2_gemm2_try_launch(const ua2_linear_args& a, hipStream_t s);

// ---- fragments ----------------------------------------------------------------------------

template <int DT> struct AFrag;
template <> struct AFrag<UA2_BF16> {
  u32x4 v;
  __device__ __forceinline__ void set(const float (&f)[8]) {
    v[0] = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
    v[1] = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
    v[2] = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
    v[3] = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
  }
  __device__ __forceinline__ void mma(const u32x4& w, f32x4& acc) const {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, w), acc, 0,
                                                  0, 0);
  }
};
template <> struct AFrag<UA2_F32> {
  f32x4 v;
  __device__ __forceinline__ void set(const float (&f)[4]) {
    v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
  }
  __device__ __forceinline__ void mma(const u32x4& w, f32x4& acc) const {
    const f32x4 b = __builtin_bit_cast(f32x4, w);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[e], b[e], acc, 0, 0, 0);
  }
};

// ---- A-operand producers (row m, K offset k0, EPL consecutive values) -----------------------

template <int EPL>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&f)[EPL]) {
#pragma unroll
  for (int q = 0; q < EPL / 4; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
    f[4 * q + 0] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
  }
}


// Fixed-order accumulators: every code path that reduces an activation row uses these, so a row's
// statistics are bit-identical whatever the batch around it (hipcc contracts a*b+c differently from one
// loop shape to another when left to itself).
__device__ __forceinline__ float sumsq4(float acc, const float4& t) {
  acc = __fmaf_rn(t.x, t.x, acc); acc = __fmaf_rn(t.y, t.y, acc);
  acc = __fmaf_rn(t.z, t.z, acc); acc = __fmaf_rn(t.w, t.w, acc);
  return acc;
}
__device__ __forceinline__ float sum4(float acc, const float4& t) {
  acc = __fadd_rn(acc, t.x); acc = __fadd_rn(acc, t.y); acc = __fadd_rn(acc, t.z); acc = __fadd_rn(acc, t.w);
  return acc;
}

// ---- normalisation flavours of UA2_PRO_NORM -----------------------------------------------------
struct NormStat { float mean, rstd; };
__device__ __forceinline__ NormStat norm_stat(const ua2_linear_args& a, float sum, float sumsq) {
  NormStat s;
  const float ms = sumsq / (float)a.K;
  if (a.norm_kind == UA2_NORM_LAYERNORM) {
    s.mean = sum / (float)a.K;
    s.rstd = 1.0f / sqrtf(fmaxf(__fsub_rn(ms, __fmul_rn(s.mean, s.mean)), 0.f) + a.eps);
  } else {
    s.mean = 0.f;
    s.rstd = 1.0f / sqrtf(ms + a.eps);     // torch.rsqrt(mean(x*x) + eps)  (lit :886-887; Moshi: eps + mean, same sum)
  }
  return s;
}
__device__ __forceinline__ float norm_apply(const ua2_linear_args& a, float x, float w, float b, const NormStat& s) {
  if (a.norm_kind == UA2_NORM_RMS_LIT) return __fmul_rn(__fmul_rn(x, s.rstd), w);      // (x*rstd)*w
  if (a.norm_kind == UA2_NORM_RMS_MOSHI) return __fmul_rn(x, __fmul_rn(w, s.rstd));    // x*(alpha*rstd)
  return __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(x, s.mean), s.rstd), w), b);                    // layer norm
}

// ---- epilogues (thread = one (row, col) of the 16 x 16 output tile; v[t] = reduced sums) ----
// Split in two so the decode kernel can issue the epilogue's global loads (residual, position,
// RoPE table entries, page id, forbid_prefix) at kernel entry, in the shadow of the weight stream,
// instead of as dependent loads after the reduction.
struct EpiPre {
  float resid = 0.f, cs = 0.f, sn = 0.f;
  float bias = 0.f, bias1 = 0.f;
  float rstd = 1.f, nw = 1.f;        // UA2_PRO_SCALED: the row's scale; y_norm_w hand-over: the following norm's weight of this column
  int pos = 0, page = 0, forbid = 0;
};

// ---- scaled-norm hand-over (UA2_PRO_SCALED, include/ua2hip.h) -----------------------------------------------------
// The one summation tree every producer uses for a row's sum of squares over a 16-column tile (lanes of one 16-lane
// group hold the 16 columns): butterfly xor 1, 2, 4, 8.  fp add is commutative, so every lane of the group ends with
// the same bits, and a producer that holds several columns per thread (ua2_misc.hip) reproduces the tree in registers.
// The first level is a FUSED multiply-add, spelled out: lane c holds fma(v_c, v_c, RN(v_{c^1}^2)) (so the two lanes of a pair
// differ in the last bit; the tile's value is what lane 0 — an even column — ends with).  Written as RN(v^2) + shfl(RN(v^2))
// hipcc contracted it to exactly this on its own (-ffp-contract=fast applies to __fmul_rn / __fadd_rn too: they are plain
// operations), while a producer holding several columns per thread got separate roundings for the same source text: the
// explicit form is what keeps the three spellings of the tree (here, ua2_misc.hip, the tiled GEMM's staged epilogue)
// on the same bits by construction instead of by the compiler's choice of the day.
__device__ __forceinline__ float ssq_tile16(float v) {
  float s = __fmaf_rn(v, v, __shfl_xor(__fmul_rn(v, v), 1));
  s = __fadd_rn(s, __shfl_xor(s, 2));
  s = __fadd_rn(s, __shfl_xor(s, 4));
  s = __fadd_rn(s, __shfl_xor(s, 8));
  return s;
}
// Row scale of a UA2_PRO_SCALED launch from the producer's partials, in two halves so that a kernel can REQUEST the partials
// before its weight burst (tiny L2 hits: they retire first) and REDUCE them after issuing it — a wave's loads retire in
// order, so partials requested behind the burst make the reduction wait for the last weight fragment, and the MFMA loop
// that follows then starts on a drained pipeline instead of on the first fragment (+4-6 us per launch, measured).
// One order for every kernel: lane c of the row's 16-lane group adds partials c, c + 16, ... ascending, the 16 chains meet
// in the butterfly xor 1, 2, 4, 8 — a row's scale does not depend on the row count or the kernel.  K <= 4096.
// NPL = partials per lane the caller holds (>= ceil(K / 256); 16 covers every K): fewer registers where K is a compile-time fact
template <int NPL>
__device__ __forceinline__ void scaled_ssq_request(const ua2_linear_args& a, int mr, int c, float (&v)[NPL]) {
  const int nparts = a.K >> 4;
  const float* p = a.x_ssq + (size_t)min(max(mr, 0), a.M - 1) * nparts;
#pragma unroll
  for (int i = 0; i < NPL; ++i) v[i] = p[min(c + 16 * i, nparts - 1)];   // clamped, not predicated: a load under a lane predicate ends in a wait per load
}
template <int NPL>
__device__ __forceinline__ float scaled_rstd_reduce(const ua2_linear_args& a, int c, const float (&v)[NPL]) {
  const int nparts = a.K >> 4;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) s = __fadd_rn(s, (c + 16 * i < nparts) ? v[i] : 0.f);   // + 0.f past the end: exact (and the same bits whatever NPL >= the need)
  s = __fadd_rn(s, __shfl_xor(s, 1));
  s = __fadd_rn(s, __shfl_xor(s, 2));
  s = __fadd_rn(s, __shfl_xor(s, 4));
  s = __fadd_rn(s, __shfl_xor(s, 8));
  return 1.0f / sqrtf(s / (float)a.K + a.eps);            // torch.rsqrt(mean(x*x) + eps), as norm_stat
}
// both halves at once, for rows [m_first, m_first + nrows) into rstd_lds (kernels without a burst to hide behind)
__device__ __forceinline__ void scaled_rstd_rows(const ua2_linear_args& a, int m_first, int nrows, float* rstd_lds, int tid, int nthreads) {
  for (int r = tid >> 4; r < nrows; r += nthreads >> 4) {
    float v[16];
    scaled_ssq_request(a, m_first + r, tid & 15, v);
    const float rs = scaled_rstd_reduce(a, tid & 15, v);
    if ((tid & 15) == 0) rstd_lds[r] = rs;
  }
}

// The same statistic for ONE row by the 16 lanes of its group (the tiled kernel's epilogue: no LDS to spare there).
__device__ __forceinline__ float scaled_rstd_row(const ua2_linear_args& a, int mr, int c) {
  float v[16];
  scaled_ssq_request(a, mr, c, v);
  return scaled_rstd_reduce(a, c, v);
}

// table row of the paged KV cache for matrix row mr: row_seq == NULL means "row r is sequence r"
__device__ __forceinline__ int kv_table_row(const ua2_linear_args& a, int mr) { return a.row_seq ? a.row_seq[mr] : mr; }

// stage A: loads that depend on nothing (issue BEFORE the weight burst)
template <int DT, int EPI>
__device__ __forceinline__ void epilogue_prefetch_a(const ua2_linear_args& a, int tile0, int row, int col, EpiPre& p, int m0) {
  const int mr = m0 + row;
  if (mr >= a.M) return;
  const int n = tile0 * 16 + col;
  if constexpr (EPI == UA2_EPI_STORE) {
    if (a.part_max && a.forbid) p.forbid = a.forbid[mr];
  } else if constexpr (EPI == UA2_EPI_RESIDUAL) {
    if (n < a.N) p.resid = a.resid[(size_t)mr * a.ldr + n];
  } else if constexpr (EPI == UA2_EPI_QKV_ROPE) {
    p.pos = a.row_pos[mr];
    p.page = kv_table_row(a, mr);   // table row for now; resolved to a page id in stage B
  }
}
// stage B: loads that depend on stage A (issue AFTER the weight burst: they queue behind it and are
// only needed by the epilogue)
template <int DT, int EPI>
__device__ __forceinline__ void epilogue_prefetch_b(const ua2_linear_args& a, int tile0, int row, int col, EpiPre& p, int m0) {
  if constexpr (EPI == UA2_EPI_STORE || EPI == UA2_EPI_RESIDUAL)
    if (a.y_norm_w && tile0 * 16 + col < a.N) p.nw = a.y_norm_w[tile0 * 16 + col];    // per output column: here, where the tile is always the real one
  if (a.bias) {                                    // nn.Linear bias: per output column, any epilogue
    const int n = tile0 * 16 + col;
    if (n < a.N) {
      p.bias = a.bias[n];
      if constexpr (EPI == UA2_EPI_SWIGLU) p.bias1 = a.bias1 ? a.bias1[n] : 0.f;
    }
  }
  if constexpr (EPI == UA2_EPI_QKV_ROPE) {
    const int mr = m0 + row;
    if (mr >= a.M) return;
    const int hs = a.kv.head_size, half = hs / 2;
    const int n0 = tile0 * 16;
    const int h = n0 / hs, r = (n0 - h * hs) / 16;
    const int d = (a.rope_mode == UA2_ROPE_INTERLEAVED) ? (r * 16 + col) / 2 : r * 8 + (col & 7);   // table column
    if (h < a.kv.n_head + a.kv.n_kv && a.rope_mode != UA2_ROPE_NONE) {
      p.cs = a.rope_cos[(size_t)p.pos * half + d];
      p.sn = a.rope_sin[(size_t)p.pos * half + d];
    }
    if (h >= a.kv.n_head) p.page = a.kv.page_table[(size_t)p.page * a.kv.max_pages + ua2_page_slot(a.kv, p.pos)];
  }
}
template <int DT, int EPI>
__device__ __forceinline__ void epilogue_prefetch(const ua2_linear_args& a, int tile0, int row, int col, EpiPre& p, int m0) {
  epilogue_prefetch_a<DT, EPI>(a, tile0, row, col, p, m0);
  epilogue_prefetch_b<DT, EPI>(a, tile0, row, col, p, m0);
}

// Producer half of the hand-over: `out` = this thread's fp32 result (row mr, column n).  16-lane shuffles: every lane of the
// row group calls it (columns >= N and rows >= M contribute zeros / store nothing).
template <int DT>
__device__ __forceinline__ void handover_emit(const ua2_linear_args& a, float out, float nw, int mr, int n, int tile, int col, bool rvalid) {
  const bool live = rvalid && n < a.N;
  const float s = ssq_tile16(live ? out : 0.f);
  if (!live) return;
  if (a.y_ssq && col == 0) a.y_ssq[(size_t)mr * ((a.N + 15) >> 4) + tile] = s;
  const float h = __fmul_rn(out, nw);
  if (a.y_h) store_elem<DT>(a.y_h, (size_t)mr * a.ldh + n, h);
  if (a.y_packed) store_packed_operand<DT>(a.y_packed, mr, n, a.N / Elem<DT>::KC, h);
}

// The activations of the SWIGLU / GELU epilogues: one definition for every kernel and epilogue form (per-element and
// staged), so the same value goes through the same instructions wherever it is computed.
__device__ __forceinline__ float ua2_act_glu(const ua2_linear_args& a, float v0, float v1) {
  if (a.act_kind == UA2_GATE_SIGMOID_SECOND)     // x-transformers GLU: x * sigmoid(gate), x = first half (w0), gate = second (w1)
    return __fmul_rn(v0, 1.0f / (1.0f + expf(-v1)));
  const float gte = v0;
  const float sg = gte / (1.0f + expf(-gte));    // F.silu, lit_model.py:594
  return __fmul_rn(sg, v1);
}
__device__ __forceinline__ float ua2_act_gelu(const ua2_linear_args& a, float x) {
  if (a.act_kind == UA2_GELU_TANH) {             // F.gelu(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
  }
  return __fmul_rn(__fmul_rn(0.5f, x), __fadd_rn(1.0f, erff(__fmul_rn(x, 0.70710678118654752440f))));
}

// NOTE: uses 16-lane shuffles: call with all 256 epilogue threads.
// HO = false compiles the hand-over emission out: the tiled kernel's STORE / RESIDUAL instantiations sit at the register
// limit (chain + total accumulators), and with the emission's shuffles in their epilogue the allocator moved the totals
// into scratch for the whole kernel (272 B/lane, the lm_head launch 222 -> 510 us) — the launcher picks the HO = true
// instantiation only for launches that hand over.
template <int DT, int EPI, int NT, bool HO = true>
__device__ __forceinline__ void linear_epilogue(const ua2_linear_args& a, const float (&vin)[NT], const int (&tile)[NT],
                                                int row, int col, const EpiPre& p, int m0, int rows) {
  const int mr = m0 + row;
  const bool rvalid = row < rows;
  float v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = vin[t];
  if (a.prologue == UA2_PRO_SCALED) {              // y = rstd * (bf16(x (.) w) W^T): one rounding on the fp32 sum
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = __fmul_rn(v[t], p.rstd);
  }
  if (a.bias) {                                    // absent (every Linear of the LM): the sums pass through untouched
    v[0] = __fadd_rn(v[0], p.bias);
    if constexpr (NT == 2) v[1] = __fadd_rn(v[1], p.bias1);
  }

  if constexpr (EPI == UA2_EPI_STORE) {
    const int n = tile[0] * 16 + col;
    if (rvalid && n < a.N && a.y) a.y[(size_t)mr * a.ldy + n] = v[0];
    if constexpr (HO) if (a.y_norm_w) handover_emit<DT>(a, v[0], p.nw, mr, n, tile[0], col, rvalid);
    if (a.part_max) {
      float bv = (n < a.N && n >= p.forbid) ? v[0] : -INFINITY;
      int bi = n;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {  // 16-lane groups; ties -> lowest index
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (col == 0 && rvalid) {
        const int nb = (a.N + 15) / 16;
        a.part_max[(size_t)mr * nb + tile[0]] = bv;
        a.part_idx[(size_t)mr * nb + tile[0]] = bi;
      }
    }
  } else if constexpr (EPI == UA2_EPI_RESIDUAL) {
    const int n = tile[0] * 16 + col;
    const float out = __fadd_rn(a.out_scale ? __fmul_rn(a.out_scale[min(n, a.N - 1)], v[0]) : v[0], p.resid);
    if (rvalid && n < a.N) a.y[(size_t)mr * a.ldy + n] = out;
    if constexpr (HO) if (a.y_norm_w) handover_emit<DT>(a, out, p.nw, mr, n, tile[0], col, rvalid);
  } else if constexpr (EPI == UA2_EPI_SWIGLU) {
    const int n = tile[0] * 16 + col;
    if (rvalid && n < a.N) {
      const float out = ua2_act_glu(a, v[0], v[1]);
      if (a.y) a.y[(size_t)mr * a.ldy + n] = out;
      if (a.y_packed) store_packed_operand<DT>(a.y_packed, mr, n, a.N / Elem<DT>::KC, out);
    }
  } else if constexpr (EPI == UA2_EPI_GELU) {
    const int n = tile[0] * 16 + col;
    if (rvalid && n < a.N) {
      const float out = ua2_act_gelu(a, v[0]);
      if (a.y) a.y[(size_t)mr * a.ldy + n] = out;
      if (a.y_packed) store_packed_operand<DT>(a.y_packed, mr, n, a.N / Elem<DT>::KC, out);
    }
  } else {  // UA2_EPI_QKV_ROPE — weight rows were permuted at pack time (ua2_pack_linear rope_head_size):
    // tile r of a head holds dims [8r, 8r+8) in columns 0-7 and their rotation partners
    // [hs/2+8r, hs/2+8r+8) in columns 8-15, so the half-split rotation closes inside one tile.
    const float other8 = __shfl_xor(v[0], 8);  // half-split partner column, same row (all 256 threads participate)
    const float other1 = __shfl_xor(v[0], 1);  // interleaved partner
    if (!rvalid) return;
    const int hs = a.kv.head_size, half = hs / 2;
    const int n0 = tile[0] * 16;
    const int h = n0 / hs, r = (n0 - h * hs) / 16;
    const bool rot = h < a.kv.n_head + a.kv.n_kv && a.rope_mode != UA2_ROPE_NONE;
    float out = v[0];
    int dd;
    if (a.rope_mode == UA2_ROPE_HALF_SPLIT) {
      const bool lo_half = col < 8;
      const int d = r * 8 + (col & 7);           // dim in [0, half)
      const float x1 = lo_half ? v[0] : other8;  // x[d]
      const float x2 = lo_half ? other8 : v[0];  // x[d + half]
      // roped = x*cos + rotate_half(x)*sin  (lit_model.py:795-806), products rounded separately
      if (rot) out = lo_half ? __fadd_rn(__fmul_rn(x1, p.cs), __fmul_rn(-x2, p.sn)) : __fadd_rn(__fmul_rn(x2, p.cs), __fmul_rn(x1, p.sn));
      dd = lo_half ? d : d + half;
    } else {
      dd = r * 16 + col;                         // natural order
      const bool even = (col & 1) == 0;
      const float xr = even ? v[0] : other1, xi = even ? other1 : v[0];
      // qor = qr*rotr - qi*roti ; qoi = qr*roti + qi*rotr   (rope.py:58-62)
      if (rot) out = even ? __fsub_rn(__fmul_rn(xr, p.cs), __fmul_rn(xi, p.sn)) : __fadd_rn(__fmul_rn(xr, p.sn), __fmul_rn(xi, p.cs));
    }
    if (h < a.kv.n_head) {
      a.q_out[(size_t)mr * a.kv.n_head * hs + (size_t)h * hs + dd] = out;
    } else {
      const bool is_k = h < a.kv.n_head + a.kv.n_kv;
      const int kvh = is_k ? h - a.kv.n_head : h - a.kv.n_head - a.kv.n_kv;
      const size_t base = (((size_t)p.page * a.kv.n_kv + kvh) * UA2_PAGE + (p.pos % UA2_PAGE)) * hs;
      store_elem<DT>(is_k ? a.kv.k_pool : a.kv.v_pool, base + dd, out);
    }
  }
}
