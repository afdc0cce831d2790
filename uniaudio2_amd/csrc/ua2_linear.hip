// Skinny-M weight-streaming GEMM for the decode loop:  Y[M,N] = epilogue( prologue(X)[M,K] * W[N,K]^T )
//
// Replaces the nn.Linear call sites of the reference's decode frame together with the op
// right before and right after each of them (include/ua2hip.h
// lists the file:line of every fused piece).
//
// MI355X design:
//   * the weight is the only HBM stream that matters (M <= 16 per workgroup row-tile), so it is
//     pre-tiled into MFMA B-fragment order (ua2_pack_linear): every wave load is one contiguous
//     1 KiB burst (64 lanes x 16 B), issued non-temporal straight into VGPRs (no LDS round trip —
//     the operand is streamed once and not shared between waves);
//   * one workgroup = 8 waves = one 16-column output tile (two tiles for the fused SwiGLU / RoPE
//     epilogues); the 8 waves split K, keep 2 x UB KiB of weight in flight each, and meet in
//     LDS for a fixed-order reduction (deterministic, batch-size independent);
//   * M rows ride in the 16-row A operand of mfma_f32_16x16x32_bf16 (bf16 operands, fp32
//     accumulate) or mfma_f32_16x16x4_f32 (exact fp32): the matrix pipe is idle-cheap at this
//     shape and the same instruction stream serves M = 1 and M = 16, so a row's result does not
//     depend on how many other rows are in flight;
//   * RMSNorm / attention-merge prologues and residual / SwiGLU / RoPE+KV-append / arg-max
//     epilogues run inside the kernel: activations never round-trip HBM in bf16.
#include "ua2_common.h"
#include "ua2_linear_common.h"

namespace {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * UA2_WAVE;
constexpr int UB = 4;  // chunks per software-pipelined batch

// ---- packing ------------------------------------------------------------------------------

template <int SRC, int DST>
__global__ void pack_kernel(const void* __restrict__ src, void* __restrict__ out, int transposed, int64_t N, int64_t K,
                            int64_t total, int rope_hs) {
  constexpr int KC = Elem<DST>::KC, EPL = Elem<DST>::EPL;
  const int64_t nchunks = (K + KC - 1) / KC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx % EPL);
    const int lane = (int)((idx / EPL) % 64);
    const int64_t chunk = (idx / (EPL * 64)) % nchunks;
    const int64_t tile = idx / (EPL * 64 * nchunks);
    int64_t n = tile * 16 + (lane & 15);
    if (rope_hs > 0) {  // packed column -> source row: [8r, 8r+8) then [hs/2+8r, hs/2+8r+8) per tile r of a head
      const int64_t h = n / rope_hs, within = n - h * rope_hs;
      const int64_t r = within / 16, c = within % 16;
      n = h * rope_hs + (c < 8 ? r * 8 + c : rope_hs / 2 + r * 8 + (c - 8));
    }
    const int64_t k = chunk * KC + (lane >> 4) * EPL + e;
    float v = 0.f;
    if (n < N && k < K) v = load_elem<SRC>(src, transposed ? (size_t)(k * N + n) : (size_t)(n * K + k));
    store_elem<DST>(out, (size_t)idx, v);
  }
}

template <int DT, int PRO>
__device__ __forceinline__ void make_a(const ua2_linear_args& a, int m, bool valid, int k0, const NormStat& st,
                                       AFrag<DT>& out) {
  constexpr int EPL = Elem<DT>::EPL;
  float f[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) f[e] = 0.f;
  if (valid && k0 < a.K) {
    if constexpr (PRO == UA2_PRO_CAST) {
      load_row<EPL>(a.x + (size_t)m * a.ldx + k0, f);
    } else if constexpr (PRO == UA2_PRO_NORM) {
      float w[EPL], b[EPL];
      load_row<EPL>(a.x + (size_t)m * a.ldx + k0, f);
      load_row<EPL>(a.norm_w + k0, w);
#pragma unroll
      for (int e = 0; e < EPL; ++e) b[e] = 0.f;
      if (a.norm_kind == UA2_NORM_LAYERNORM) load_row<EPL>(a.norm_b + k0, b);
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] = norm_apply(a, f[e], w[e], b[e], st);
    } else {  // UA2_PRO_ATTN: merge the per-page partials of head h
      const int hs = a.kv.head_size, mp = a.kv.max_pages;
      const int h = k0 / hs, d = k0 - h * hs;
      const int nsp = a.row_pos[m] / UA2_PAGE + 1;
      const float* ml = a.attn_ml + ((size_t)m * a.kv.n_head + h) * mp * 2;
      const float* po = a.attn_o + (((size_t)m * a.kv.n_head + h) * mp) * hs + d;
      float mx = -INFINITY;
      for (int s = 0; s < nsp; ++s) mx = fmaxf(mx, ml[2 * s]);
      float den = 0.f;
      for (int s = 0; s < nsp; ++s) {
        const float wgt = expf(ml[2 * s] - mx);
        den += wgt * ml[2 * s + 1];
        float t[EPL];
        load_row<EPL>(po + (size_t)s * hs, t);
#pragma unroll
        for (int e = 0; e < EPL; ++e) f[e] += wgt * t[e];
      }
      const float inv = 1.0f / den;
#pragma unroll
      for (int e = 0; e < EPL; ++e) f[e] *= inv;
    }
  }
  out.set(f);
}

// ---- the kernel ---------------------------------------------------------------------------

template <int DT, int PRO, int EPI>
__global__ __launch_bounds__(kThreads) void linear_kernel(const ua2_linear_args a) {
  constexpr int KC = Elem<DT>::KC, EPL = Elem<DT>::EPL;
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  __shared__ float red[kWaves][NT][256];
  __shared__ float ssq[kWaves][16], ssum[kWaves][16];
  __shared__ float rstd_s[16], mean_s[16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int m = blockIdx.y * 16 + i;
  const bool mvalid = m < a.M;

  const int nchunks = (a.K + KC - 1) / KC;
  int tile[NT];
  const u32x4* wp[NT];
  if constexpr (EPI == UA2_EPI_SWIGLU) {
    tile[0] = tile[1] = blockIdx.x;
    wp[0] = reinterpret_cast<const u32x4*>(a.w0) + (size_t)tile[0] * nchunks * 64 + lane;
    wp[1] = reinterpret_cast<const u32x4*>(a.w1) + (size_t)tile[0] * nchunks * 64 + lane;
  } else {
    tile[0] = blockIdx.x;
    wp[0] = reinterpret_cast<const u32x4*>(a.w0) + (size_t)tile[0] * nchunks * 64 + lane;
  }
  const int c0 = (wave * nchunks) / kWaves, c1 = ((wave + 1) * nchunks) / kWaves;

  // first weight batch goes out before anything else so HBM latency overlaps the prologue
  u32x4 wf[NT][UB];
  const bool full0 = c0 + UB <= c1;
  if (full0) {
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(c0 + u) * 64);
  }

  NormStat nst{0.f, 1.f};
  if constexpr (PRO == UA2_PRO_NORM) {
    float ss = 0.f, sm = 0.f;
    if (mvalid) {
      for (int c = c0; c < c1; ++c) {
        const int k0 = c * KC + g * EPL;
        if (k0 < a.K) {
          float f[EPL];
          load_row<EPL>(a.x + (size_t)m * a.ldx + k0, f);
#pragma unroll
          for (int e = 0; e < EPL; ++e) { ss += f[e] * f[e]; sm += f[e]; }
        }
      }
    }
    ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32);
    sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
    if (g == 0) { ssq[wave][i] = ss; ssum[wave][i] = sm; }
    __syncthreads();
    if (tid < 16) {
      float t = 0.f, u = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) { t += ssq[w][tid]; u += ssum[w][tid]; }
      const NormStat st = norm_stat(a, u, t);
      rstd_s[tid] = st.rstd; mean_s[tid] = st.mean;
    }
    __syncthreads();
    nst.rstd = rstd_s[i]; nst.mean = mean_s[i];
  }

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  int c = c0;
  if (full0) {
    // steady state: A for batch b, prefetch W for batch b+1, then the MFMAs of batch b
    for (; c + UB <= c1; c += UB) {
      AFrag<DT> af[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) make_a<DT, PRO>(a, m, mvalid, (c + u) * KC + g * EPL, nst, af[u]);
      u32x4 wn[NT][UB];
      const bool more = c + 2 * UB <= c1;
      if (more) {
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) wn[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)(c + UB + u) * 64);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) af[u].mma(wf[t][u], acc[t]);
      if (more) {
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[t][u] = wn[t][u];
      }
    }
  }
  for (; c < c1; ++c) {  // remainder chunks (K not a multiple of 8*UB*KC)
    AFrag<DT> af;
    make_a<DT, PRO>(a, m, mvalid, c * KC + g * EPL, nst, af);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4 w = __builtin_nontemporal_load(wp[t] + (size_t)c * 64);
      af.mma(w, acc[t]);
    }
  }

  // fixed-order cross-wave reduction through LDS
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][lane * 4]) = acc[t];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, col = tid & 15;
  const int src = (((row >> 2) << 4) + col) * 4 + (row & 3);  // C/D layout: lane=(row/4)*16+col, reg=row%4
  float v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) s += red[w][t][src];
    v[t] = s;
  }
  EpiPre pre;
  const int m0e = blockIdx.y * 16;
  epilogue_prefetch<DT, EPI>(a, tile[0], row, col, pre, m0e);
  linear_epilogue<DT, EPI, NT>(a, v, tile, row, col, pre, m0e, min(16, a.M - m0e));
}

template <int DT, int PRO>
int launch_epi(const ua2_linear_args& a, hipStream_t s) {
  const int ntiles = ua2_ceil_div(a.N, 16);
  const dim3 block(kThreads);
  const int mtiles = ua2_ceil_div(a.M, 16);
  switch (a.epilogue) {
    case UA2_EPI_STORE:
      hipLaunchKernelGGL((linear_kernel<DT, PRO, UA2_EPI_STORE>), dim3(ntiles, mtiles), block, 0, s, a);
      break;
    case UA2_EPI_RESIDUAL:
      hipLaunchKernelGGL((linear_kernel<DT, PRO, UA2_EPI_RESIDUAL>), dim3(ntiles, mtiles), block, 0, s, a);
      break;
    case UA2_EPI_SWIGLU:
      hipLaunchKernelGGL((linear_kernel<DT, PRO, UA2_EPI_SWIGLU>), dim3(ntiles, mtiles), block, 0, s, a);
      break;
    case UA2_EPI_QKV_ROPE:
      hipLaunchKernelGGL((linear_kernel<DT, PRO, UA2_EPI_QKV_ROPE>), dim3(ntiles, mtiles), block, 0, s, a);
      break;
    case UA2_EPI_GELU:
      hipLaunchKernelGGL((linear_kernel<DT, PRO, UA2_EPI_GELU>), dim3(ntiles, mtiles), block, 0, s, a);
      break;
    default:
      ua2_set_error("ua2_linear: bad epilogue %d", a.epilogue);
      return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

template <int DT>
int launch_pro(const ua2_linear_args& a, hipStream_t s) {
  switch (a.prologue) {
    case UA2_PRO_CAST: return launch_epi<DT, UA2_PRO_CAST>(a, s);
    case UA2_PRO_NORM: return launch_epi<DT, UA2_PRO_NORM>(a, s);
    case UA2_PRO_ATTN: return launch_epi<DT, UA2_PRO_ATTN>(a, s);
  }
  ua2_set_error("ua2_linear: bad prologue %d", a.prologue);
  return -1;
}

}  // namespace

static int g_force_general = 0;

int ua2_linear_launch(const ua2_linear_args& a, hipStream_t s) {
  UA2_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "ua2_linear: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  UA2_CHECK(a.w0 != nullptr, "ua2_linear: w0 is NULL");
  const int epl = a.dtype == UA2_BF16 ? 8 : 4;
  UA2_CHECK(a.K % epl == 0, "ua2_linear: K=%d must be a multiple of %d", a.K, epl);
  if (a.prologue == UA2_PRO_LOCAL_ATTN) {
    const int kc = a.dtype == UA2_BF16 ? 32 : 16;
    UA2_CHECK(a.M == 1 && a.epilogue == UA2_EPI_RESIDUAL, "ua2_linear: LOCAL_ATTN is the M == 1 O-projection only (use ua2_attn_local + CAST otherwise)");
    UA2_CHECK(a.x && a.row_pos && a.kv.k_pool && a.kv.v_pool && a.kv.page_table && a.kv.n_kv > 0 && a.kv.n_head % a.kv.n_kv == 0 &&
                  (a.kv.head_size == 32 || a.kv.head_size == 64 || a.kv.head_size == 128) &&
                  a.kv.n_head % (128 / a.kv.head_size) == 0 && a.K == a.kv.n_head * a.kv.head_size && a.K % kc == 0,
              "ua2_linear: bad LOCAL_ATTN arguments");
    UA2_CHECK(a.resid != nullptr && a.y != nullptr, "ua2_linear: RESIDUAL needs resid, y");
    const int rc = ua2_gemv_try_launch(a, s);
    UA2_CHECK(rc <= 0, "ua2_linear: LOCAL_ATTN problem outside the decode kernel's range");
    return rc;
  }
  if (a.x_packed) {   // operand handed over in fragment order by its producer: only the many-row kernels read it
    UA2_CHECK(a.prologue == UA2_PRO_CAST && g_force_general != 1 && g_force_general != 2, "ua2_linear: x_packed needs PRO_CAST and the many-row kernels");
    const int rc = ua2_gemm_try_launch(a, s, 3);
    UA2_CHECK(rc <= 0, "ua2_linear: x_packed launch not applicable");
    return rc;
  }
  if (a.prologue != UA2_PRO_ATTN) {
    UA2_CHECK(a.x != nullptr && a.ldx % 4 == 0, "ua2_linear: x NULL or ldx %% 4 != 0");
  } else {
    UA2_CHECK(a.attn_o && a.attn_ml && a.row_pos && a.kv.head_size % 16 == 0 &&
                  a.K == a.kv.n_head * a.kv.head_size,
              "ua2_linear: bad ATTN prologue arguments");
  }
  if (a.prologue == UA2_PRO_NORM)
    UA2_CHECK(a.norm_w != nullptr && (a.norm_kind != UA2_NORM_LAYERNORM || a.norm_b != nullptr) && a.norm_kind >= 0 && a.norm_kind <= 2,
              "ua2_linear: norm_w / norm_b / norm_kind invalid");
  if (a.epilogue == UA2_EPI_GELU) UA2_CHECK(a.y != nullptr, "ua2_linear: GELU needs y");
  if (a.epilogue == UA2_EPI_SWIGLU) {
    UA2_CHECK(a.w1 != nullptr && (a.y != nullptr || a.y_packed != nullptr), "ua2_linear: SWIGLU needs w1 and y or y_packed");
    UA2_CHECK(!a.y_packed || a.N % (a.dtype == UA2_BF16 ? 32 : 16) == 0, "ua2_linear: y_packed needs N %% chunk == 0");
  } else {
    UA2_CHECK(!a.y_packed, "ua2_linear: y_packed is a SWIGLU output");
  }
  if (a.epilogue == UA2_EPI_RESIDUAL) UA2_CHECK(a.resid != nullptr && a.y != nullptr, "ua2_linear: RESIDUAL needs resid, y");
  if (a.epilogue == UA2_EPI_STORE) UA2_CHECK(a.y != nullptr || a.part_max != nullptr, "ua2_linear: STORE needs y or part_max");
  if (a.epilogue == UA2_EPI_QKV_ROPE) {
    UA2_CHECK(a.kv.head_size % (a.rope_mode == UA2_ROPE_HALF_SPLIT ? 32 : 16) == 0 &&
                  a.N == (a.kv.n_head + 2 * a.kv.n_kv) * a.kv.head_size && a.rope_mode >= 0 && a.rope_mode <= 2,
              "ua2_linear: QKV_ROPE needs head_size %% 32 == 0 (16 when not half-split) and N == (n_head+2*n_kv)*head_size");
    UA2_CHECK(a.row_pos && (a.rope_mode == UA2_ROPE_NONE || (a.rope_cos && a.rope_sin)) && a.q_out && a.kv.k_pool && a.kv.v_pool &&
                  a.kv.page_table,
              "ua2_linear: QKV_ROPE pointer arguments missing");
  }
  if (a.dtype != UA2_BF16 && a.dtype != UA2_F32) {
    ua2_set_error("ua2_linear: bad dtype %d", a.dtype);
    return -1;
  }
  if (g_force_general != 1) {
    if (g_force_general != 2) {
      const int rc = ua2_gemm_try_launch(a, s, g_force_general >= 3 ? g_force_general : 0);  // many rows: packed operand, 128-row tiles
      if (rc <= 0) return rc;
    }
    const int rc = ua2_gemv_try_launch(a, s);  // decode regime: LDS-staged activations, all loads up front
    if (rc <= 0) return rc;
  }
  if (a.dtype == UA2_BF16) return launch_pro<UA2_BF16>(a, s);
  return launch_pro<UA2_F32>(a, s);
}

extern "C" int ua2_debug_force_general_linear(int on) {
  const int old = g_force_general;
  g_force_general = on;
  return old;
}

extern "C" int ua2_linear(const ua2_linear_args* a, void* stream) {
  UA2_CHECK(a != nullptr, "ua2_linear: NULL args");
  return ua2_linear_launch(*a, (hipStream_t)stream);
}

extern "C" int ua2_linear_chain_timed(const ua2_linear_args* args, int32_t n, int32_t iters, void* stream,
                                      float* ms_out) {
  UA2_CHECK(args && n > 0 && iters > 0 && ms_out, "ua2_linear_chain_timed: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  UA2_HIP(hipEventCreate(&e0));
  UA2_HIP(hipEventCreate(&e1));
  UA2_HIP(hipEventRecord(e0, s));
  for (int it = 0; it < iters; ++it)
    for (int i = 0; i < n; ++i)
      if (int rc = ua2_linear_launch(args[i], s)) return rc;
  UA2_HIP(hipEventRecord(e1, s));
  UA2_HIP(hipEventSynchronize(e1));
  UA2_HIP(hipEventElapsedTime(ms_out, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

extern "C" size_t ua2_packed_elems(int dtype, int64_t N, int64_t K) {
  const int kc = dtype == UA2_BF16 ? 32 : 16, epl = dtype == UA2_BF16 ? 8 : 4;
  return (size_t)((N + 15) / 16) * (size_t)((K + kc - 1) / kc) * 64 * epl;
}

extern "C" int ua2_pack_linear(const void* src, int src_dtype, int transposed, int64_t N, int64_t K, void* out,
                               int dtype, int rope_head_size, void* stream) {
  UA2_CHECK(src && out && N > 0 && K > 0, "ua2_pack_linear: bad arguments");
  UA2_CHECK(rope_head_size == 0 || (rope_head_size % 32 == 0 && N % rope_head_size == 0),
            "ua2_pack_linear: rope_head_size=%d must divide N and be a multiple of 32", rope_head_size);
  const int64_t total = (int64_t)ua2_packed_elems(dtype, N, K);
  const int threads = 256;
  const int blocks = (int)((total + threads - 1) / threads < 65535 * 16 ? (total + threads - 1) / threads : 65535 * 16);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == UA2_F32 && dtype == UA2_F32)
    hipLaunchKernelGGL((pack_kernel<UA2_F32, UA2_F32>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else if (src_dtype == UA2_F32 && dtype == UA2_BF16)
    hipLaunchKernelGGL((pack_kernel<UA2_F32, UA2_BF16>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else if (src_dtype == UA2_BF16 && dtype == UA2_BF16)
    hipLaunchKernelGGL((pack_kernel<UA2_BF16, UA2_BF16>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else if (src_dtype == UA2_BF16 && dtype == UA2_F32)
    hipLaunchKernelGGL((pack_kernel<UA2_BF16, UA2_F32>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else {
    ua2_set_error("ua2_pack_linear: bad dtypes %d -> %d", src_dtype, dtype);
    return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}
