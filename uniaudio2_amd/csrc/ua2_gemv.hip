// Decode-regime weight-streaming GEMM (M <= 16 rows per tile, activations small enough for LDS).
//
// Same contract and epilogues as ua2_linear.hip (include/ua2hip.h ua2_linear), restructured after
// the first rocprof pass (profiles/r1_a_*): at B = 1 every kernel of the frame was bound by
// dependent-load latency, not bandwidth.  What changed, and why it maps to CDNA4:
//   * every wave issues ALL of its weight loads (CPW x NT non-temporal 1 KiB bursts) as its first
//     instructions; nothing in the prologue is ordered before them, so up to 16 waves x 16 KiB of
//     HBM traffic per CU is in flight while the activations are prepared;
//   * the activation rows are staged once per workgroup: coalesced float4 loads by all threads,
//     RMSNorm statistics by a workgroup reduction, normalised values written to LDS in the MFMA
//     operand dtype; A fragments are then ds_read_b128 (one per chunk) instead of per-wave
//     global loads that serialised behind the weight stream;
//   * the wave count (8-16) and chunks-per-wave (CPW) are chosen by the launcher so that
//     waves x CPW tiles K exactly for the model's shapes: no tail, no predicated loads.
#include <stdlib.h>

#include <algorithm>

#include "ua2_common.h"
#include "ua2_linear_common.h"
#include "ua2_attn_local.h"

namespace {

constexpr int kMaxWaves = 16;

// UA2_PRO_LOCAL_ATTN: this wave's share of the row's short-context attention, written straight into the LDS
// operand row.  `burst` issues the wave's weight loads; it is called after the first pass's small loads are out
// (a wave's loads return in order: q / K / V must not queue behind the weight stream).
template <int DT, int HS, typename Burst>
__device__ __forceinline__ void local_attn_prologue(const ua2_linear_args& a, char* a_lds, int m, int wave, int nw, int lane,
                                                    Burst&& burst) {
  using LA = LocalAttn<DT, HS>;
  constexpr int BYTES = Elem<DT>::BYTES;
  const int pos = a.row_pos[m];
  const int page = a.kv.page_table[(size_t)kv_table_row(a, m) * a.kv.max_pages];
  const float* q_row = a.x + (size_t)m * a.ldx;
  bool streamed = false;
  for (int h0 = wave * LA::HPW; h0 < a.kv.n_head; h0 += nw * LA::HPW) {
    const int h = h0 + lane / LA::LPH, d = (lane % LA::LPH) * 2;
    LA la;
    la.issue(a.kv, q_row, page, h, d);
    if (!streamed) { burst(); streamed = true; }
    const float2 o = la.finish(pos);
    char* dst = a_lds + ((size_t)h * HS + d) * BYTES;
    if constexpr (DT == UA2_BF16) *reinterpret_cast<unsigned*>(dst) = (unsigned)f2bf(o.x) | ((unsigned)f2bf(o.y) << 16);
    else *reinterpret_cast<float2*>(dst) = o;
  }
  if (!streamed) burst();
}

// MR = multi-round: <= 8 waves per workgroup, several rounds of CPW chunks per wave, next round's
// weights prefetched (double buffer); !MR = single burst: up to 16 waves, everything up front.
template <int DT, int PRO, int EPI, int CPW, bool MR>
__device__ __forceinline__ void gemv_body(const ua2_linear_args& a, char* smem, const int a_stride, const int red_off, const int rt,
                                          const int bx, const int by) {
  constexpr int KC = Elem<DT>::KC, EPL = Elem<DT>::EPL, BYTES = Elem<DT>::BYTES;
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  char* a_lds = smem;                                        // [rows][a_stride] of T
  float* red = reinterpret_cast<float*>(smem + red_off);     // [nw][NT][256]
  float* ssq = red + kMaxWaves * NT * 256;                   // [nw][16]
  float* ssum = ssq + kMaxWaves * 16;                        // [nw][16]
  float* rstd_s = ssum + kMaxWaves * 16;                     // [16]
  float* mean_s = rstd_s + 16;                               // [16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nw = nthreads >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int m0 = by * rt;                    // rt rows per workgroup: a function of K and dtype only
  const int rows = min(rt, a.M - m0);

  const int nchunks = (a.K + KC - 1) / KC;
  int tile[NT];
  const u32x4* wp[NT];
  if constexpr (EPI == UA2_EPI_SWIGLU) {
    tile[0] = tile[1] = bx;
    wp[0] = reinterpret_cast<const u32x4*>(a.w0) + (size_t)tile[0] * nchunks * 64 + lane;
    wp[1] = reinterpret_cast<const u32x4*>(a.w1) + (size_t)tile[0] * nchunks * 64 + lane;
  } else {
    tile[0] = bx;
    wp[0] = reinterpret_cast<const u32x4*>(a.w0) + (size_t)tile[0] * nchunks * 64 + lane;
  }
  // chunk range of this wave; rounds of CPW chunks (exactly one round for the model's shapes)
  const int c0 = (wave * nchunks) / nw, c1 = ((wave + 1) * nchunks) / nw;
  const int last = max(c1 - 1, 0);

  // Issue order matters: a wave's loads return in order, so everything small that the prologue or
  // the epilogue needs goes out BEFORE the weight burst (it then completes in one L2 round trip
  // while the weights are still streaming), never behind it.
  constexpr int XPT = 2;  // float4 per thread kept in registers on the single-row fast path
  const bool one = (PRO != UA2_PRO_LOCAL_ATTN) && (PRO != UA2_PRO_SCALED) && (rows == 1) && (a.K <= XPT * 4 * nthreads);
  // UA2_PRO_SCALED: the operand row(s) arrive already rounded (RNE_bf16(x (.) w), written by the producer of x): a 16-byte
  // copy into the LDS tile — no statistics, no conversion, one barrier.  The first row's piece goes out before the weights.
  u32x4 xh0 = u32x4{0u, 0u, 0u, 0u};
  if constexpr (PRO == UA2_PRO_SCALED) {
    if (tid * 8 < a.K) xh0 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.x_h) + (size_t)m0 * a.ldh + tid * 8);
  }
  float4 xv[XPT], nv[XPT], nb[XPT];
  const bool ln = (PRO == UA2_PRO_NORM) && a.norm_kind == UA2_NORM_LAYERNORM;
  if (one) {
    const float* xr = a.x + (size_t)m0 * a.ldx;
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
      const int k = (tid + it * nthreads) * 4;
      xv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      nv[it] = make_float4(1.f, 1.f, 1.f, 1.f);
      nb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < a.K) {
        xv[it] = *reinterpret_cast<const float4*>(xr + k);
        if constexpr (PRO == UA2_PRO_NORM) {
          nv[it] = *reinterpret_cast<const float4*>(a.norm_w + k);
          if (ln) nb[it] = *reinterpret_cast<const float4*>(a.norm_b + k);
        }
      }
    }
  }
  EpiPre pre;
  if (tid < 256) epilogue_prefetch_a<DT, EPI>(a, tile[0], tid >> 4, tid & 15, pre, m0);
  float ssqv[16];                                        // UA2_PRO_SCALED: the row's sum-of-squares partials, requested before the burst
  if constexpr (PRO == UA2_PRO_SCALED) {
    if (tid < 256) scaled_ssq_request(a, m0 + min(tid >> 4, rows - 1), tid & 15, ssqv);
  }

  u32x4 wf[NT][CPW];
  auto burst = [&]() {
#pragma unroll
    for (int u = 0; u < CPW; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)min(c0 + u, last) * 64);
  };
  if constexpr (PRO == UA2_PRO_LOCAL_ATTN) {
    switch (a.kv.head_size) {
      case 32: local_attn_prologue<DT, 32>(a, a_lds, m0, wave, nw, lane, burst); break;
      case 64: local_attn_prologue<DT, 64>(a, a_lds, m0, wave, nw, lane, burst); break;
      default: local_attn_prologue<DT, 128>(a, a_lds, m0, wave, nw, lane, burst); break;
    }
  } else {
    burst();
  }
  if (tid < 256) epilogue_prefetch_b<DT, EPI>(a, tile[0], tid >> 4, tid & 15, pre, m0);

  // ---- stage the activation rows into LDS (operand dtype) ----
  auto put = [&](int r0, int k, float4 t) {
    char* dst = a_lds + ((size_t)r0 * a_stride + k) * BYTES;
    if constexpr (DT == UA2_BF16) {
      uint2 p;
      p.x = (unsigned)f2bf(t.x) | ((unsigned)f2bf(t.y) << 16);
      p.y = (unsigned)f2bf(t.z) | ((unsigned)f2bf(t.w) << 16);
      *reinterpret_cast<uint2*>(dst) = p;
    } else {
      *reinterpret_cast<float4*>(dst) = t;
    }
  };
  if constexpr (PRO == UA2_PRO_LOCAL_ATTN) {
    // operand row already in LDS
  } else if constexpr (PRO == UA2_PRO_SCALED) {
    static_assert(DT == UA2_BF16 || PRO != UA2_PRO_SCALED, "scaled hand-over is a bf16 contract");
    const int kpad = nchunks * KC;                               // K % 32 == 0 (checked by the launcher): kpad == K
    for (int r0 = 0; r0 < rows; ++r0) {
      const unsigned short* xr = reinterpret_cast<const unsigned short*>(a.x_h) + (size_t)(m0 + r0) * a.ldh;
      for (int k = tid * 8; k < kpad; k += nthreads * 8) {
        const u32x4 v = (r0 == 0 && k == tid * 8) ? xh0 : *reinterpret_cast<const u32x4*>(xr + k);
        *reinterpret_cast<u32x4*>(a_lds + ((size_t)r0 * a_stride + k) * BYTES) = v;
      }
    }
  } else if (one) {
    NormStat st{0.f, 1.f};
    if constexpr (PRO == UA2_PRO_NORM) {
      float ss = 0.f, sm = 0.f;
#pragma unroll
      for (int it = 0; it < XPT; ++it) {
        ss = sumsq4(ss, xv[it]);
        sm = sum4(sm, xv[it]);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { ss += __shfl_xor(ss, o); sm += __shfl_xor(sm, o); }
      if (lane == 0) { ssq[wave * 16] = ss; ssum[wave * 16] = sm; }
      __syncthreads();
      float t = 0.f, u = 0.f;
      for (int w = 0; w < nw; ++w) { t += ssq[w * 16]; u += ssum[w * 16]; }   // every thread, same order: no second barrier
      st = norm_stat(a, u, t);
    }
#pragma unroll
    for (int it = 0; it < XPT; ++it) {
      const int k = (tid + it * nthreads) * 4;
      if (k < nchunks * KC) {                             // also zero-fills the K padding of the last chunk
        float4 t = xv[it];
        if constexpr (PRO == UA2_PRO_NORM) {
          if (k < a.K) {
            t.x = norm_apply(a, t.x, nv[it].x, nb[it].x, st);
            t.y = norm_apply(a, t.y, nv[it].y, nb[it].y, st);
            t.z = norm_apply(a, t.z, nv[it].z, nb[it].z, st);
            t.w = norm_apply(a, t.w, nv[it].w, nb[it].w, st);
          }
        }
        put(0, k, t);
      }
    }
  } else {
    if constexpr (PRO == UA2_PRO_NORM) {
      for (int r0 = 0; r0 < rows; ++r0) {  // pass 1: sum and sum of squares per row
        const float* xr = a.x + (size_t)(m0 + r0) * a.ldx;
        float ss = 0.f, sm = 0.f;
        for (int k = tid * 4; k < a.K; k += nthreads * 4) {
          const float4 t = *reinterpret_cast<const float4*>(xr + k);
          ss = sumsq4(ss, t);
          sm = sum4(sm, t);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { ss += __shfl_xor(ss, o); sm += __shfl_xor(sm, o); }
        if (lane == 0) { ssq[wave * 16 + r0] = ss; ssum[wave * 16 + r0] = sm; }
      }
      __syncthreads();
      if (tid < rows) {
        float t = 0.f, u = 0.f;
        for (int w = 0; w < nw; ++w) { t += ssq[w * 16 + tid]; u += ssum[w * 16 + tid]; }
        const NormStat st = norm_stat(a, u, t);
        rstd_s[tid] = st.rstd; mean_s[tid] = st.mean;
      }
      __syncthreads();
    }
    for (int r0 = 0; r0 < rows; ++r0) {
      const float* xr = a.x + (size_t)(m0 + r0) * a.ldx;
      NormStat st{0.f, 1.f};
      if constexpr (PRO == UA2_PRO_NORM) { st.rstd = rstd_s[r0]; st.mean = mean_s[r0]; }
      for (int k = tid * 4; k < nchunks * KC; k += nthreads * 4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < a.K) {
          t = *reinterpret_cast<const float4*>(xr + k);
          if constexpr (PRO == UA2_PRO_NORM) {
            const float4 w = *reinterpret_cast<const float4*>(a.norm_w + k);
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ln) b = *reinterpret_cast<const float4*>(a.norm_b + k);
            t.x = norm_apply(a, t.x, w.x, b.x, st);
            t.y = norm_apply(a, t.y, w.y, b.y, st);
            t.z = norm_apply(a, t.z, w.z, b.z, st);
            t.w = norm_apply(a, t.w, w.w, b.w, st);
          }
        }
        put(r0, k, t);
      }
    }
  }
  if constexpr (PRO == UA2_PRO_SCALED) {
    if (tid < 256) pre.rstd = scaled_rstd_reduce(a, tid & 15, ssqv);     // requested before the burst: landed long ago
  }
  __syncthreads();

  // ---- MFMA over this wave's chunks ----
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool arow = i < rows;
  const char* abase = a_lds + ((size_t)i * a_stride + g * EPL) * BYTES;
  for (int cb = c0; cb < c1; cb += CPW) {
    // multi-round geometries: the next round's weights go out before this round's MFMAs
    u32x4 wn[NT][MR ? CPW : 1];
    const bool more = MR && (cb + CPW < c1);
    if constexpr (MR) if (more) {
#pragma unroll
      for (int u = 0; u < CPW; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) wn[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)min(cb + CPW + u, last) * 64);
    }
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      const int c = cb + u;
      AFrag<DT> af;
      u32x4 raw = u32x4{0u, 0u, 0u, 0u};
      if (arow && c < c1) raw = *reinterpret_cast<const u32x4*>(abase + (size_t)c * KC * BYTES);
      af.v = __builtin_bit_cast(decltype(af.v), raw);
#pragma unroll
      for (int t = 0; t < NT; ++t) af.mma(wf[t][u], acc[t]);
    }
    if constexpr (MR) {
      if (more) {
#pragma unroll
        for (int u = 0; u < CPW; ++u)
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[t][u] = wn[t][u];
      }
    } else if (cb + CPW < c1) {  // rare: single-burst geometry that does not tile K exactly
#pragma unroll
      for (int u = 0; u < CPW; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t][u] = __builtin_nontemporal_load(wp[t] + (size_t)min(cb + CPW + u, last) * 64);
    }
  }

  // ---- fixed-order cross-wave reduction, epilogue ----
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(&red[(wave * NT + t) * 256 + lane * 4]) = acc[t];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, col = tid & 15;
  const int src = (((row >> 2) << 4) + col) * 4 + (row & 3);
  float v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[(w * NT + t) * 256 + src];
    v[t] = s;
  }
  linear_epilogue<DT, EPI, NT>(a, v, tile, row, col, pre, m0, rows);
}

template <int DT, int PRO, int EPI, int CPW, bool MR>
__global__ __launch_bounds__(MR ? 512 : kMaxWaves * 64) void gemv_kernel(const ua2_linear_args a, const int a_stride,
                                                              const int red_off, const int rt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemv_body<DT, PRO, EPI, CPW, MR>(a, smem, a_stride, red_off, rt, blockIdx.x, blockIdx.y);
}

// ---- riders: a second, independent one-row-tile GEMV carried by the idle CUs of a launch that cannot fill the device -----------
// Why.  A CU takes in at most ~31 GB/s of an HBM stream (~64 KB in flight per CU: profiles/r6_notes.md), so a launch with fewer
// workgroups than CUs is capped below the HBM rate by the CUs it does not use: the depth decoder's down-projection (K = 8192 ->
// N = 2048: 128 workgroups of 262 KB) streams 33.5 MB in 10.6 us = 3.2 TB/s with half the device idle, 32 times per frame.
// lm_head (model_new.py:617: 788 MB, 114 us on its own) depends only on the trunk's output, not on the depth decoder
// (:629-640), and a forked graph branch replays 0.4 ms per frame SLOWER on ROCm 7.2 (measured in rounds 1 and 6).  So its column
// tiles ride along: workgroups past the host's grid each take TWO rider tiles (waves 0-7 and 8-15: the rider's eight K ranges
// each, the operand row staged once), and one slice of lm_head goes with every down-projection launch of the frame.
// Same bits as the rider's own launch (gemv_kernel<bf16, CAST, STORE, 4, MR>): the bf16 operand row, per range one MFMA chain
// over its CPWR chunks in order, the eight partial sums added in range order from zero, linear_epilogue.
template <int CPWR>
__device__ __forceinline__ void rider_body(const ua2_linear_args& a, char* smem, const int tile0, const int tile1, const int rid) {
  constexpr int DT = UA2_BF16, KC = 32, NWR = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = wave >> 3, wv = wave & 7;
  const int i = lane & 15, g = lane >> 4;
  const int rows = a.M;                                      // one row tile (the launcher checks)
  constexpr int nchunks = NWR * CPWR;                        // == K / 32 (the launcher checks)
  constexpr int a_stride = nchunks * KC + 8;
  char* a_lds = smem;
  float* red = reinterpret_cast<float*>(smem + (((size_t)rows * a_stride * 2 + 255) & ~(size_t)255));     // [2][NWR][256]
  const int tile = tile0 + 2 * rid + half;
  const bool live = tile < tile1;                            // wave-uniform
  const int tl = min(tile, tile1 - 1);
  const u32x4* wp = reinterpret_cast<const u32x4*>(a.w0) + ((size_t)tl * nchunks + wv * CPWR) * 64 + lane;
  const int et = tid & 511;                                  // epilogue thread of this half: (row, col) = (et >> 4, et & 15)
  // small loads first (a wave's loads return in order): the single-row operand piece, the epilogue's row data
  const bool one = rows == 1 && a.K <= 4 * kMaxWaves * 64;
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (one && tid * 4 < a.K) xv = *reinterpret_cast<const float4*>(a.x + tid * 4);
  EpiPre pre;
  if (et < 256) epilogue_prefetch_a<DT, UA2_EPI_STORE>(a, tl, et >> 4, et & 15, pre, 0);
  u32x4 wf[CPWR];
#pragma unroll
  for (int u = 0; u < CPWR; ++u) wf[u] = __builtin_nontemporal_load(wp + (size_t)u * 64);
  if (et < 256) epilogue_prefetch_b<DT, UA2_EPI_STORE>(a, tl, et >> 4, et & 15, pre, 0);
  auto put = [&](int r0, int k, const float4& t) {
    uint2 p;
    p.x = (unsigned)f2bf(t.x) | ((unsigned)f2bf(t.y) << 16);
    p.y = (unsigned)f2bf(t.z) | ((unsigned)f2bf(t.w) << 16);
    *reinterpret_cast<uint2*>(a_lds + ((size_t)r0 * a_stride + k) * 2) = p;
  };
  if (one) {
    if (tid * 4 < a.K) put(0, tid * 4, xv);
  } else {
    for (int r0 = 0; r0 < rows; ++r0)
      for (int k = tid * 4; k < a.K; k += kMaxWaves * 64 * 4) put(r0, k, *reinterpret_cast<const float4*>(a.x + (size_t)r0 * a.ldx + k));
  }
  __syncthreads();
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool arow = i < rows;
  const char* abase = a_lds + ((size_t)i * a_stride + g * 8) * 2 + (size_t)(wv * CPWR) * KC * 2;
#pragma unroll
  for (int u = 0; u < CPWR; ++u) {
    AFrag<DT> af;
    u32x4 raw = u32x4{0u, 0u, 0u, 0u};
    if (arow) raw = *reinterpret_cast<const u32x4*>(abase + (size_t)u * KC * 2);
    af.v = raw;
    af.mma(wf[u], acc);
  }
  *reinterpret_cast<f32x4*>(&red[((half * NWR + wv) * 256) + lane * 4]) = acc;
  __syncthreads();
  if (et >= 256 || !live) return;
  const int row = et >> 4, col = et & 15;
  const int src = (((row >> 2) << 4) + col) * 4 + (row & 3);
  float v[1];
  {
    float sacc = 0.f;
    for (int w = 0; w < NWR; ++w) sacc += red[(half * NWR + w) * 256 + src];
    v[0] = sacc;
  }
  const int tl1[1] = {tile};
  linear_epilogue<DT, UA2_EPI_STORE, 1>(a, v, tl1, row, col, pre, 0, rows);
}

// host = a single-burst 16-wave bf16 instantiation of gemv_body (one row tile); workgroups >= host_gx are riders
template <int PRO, int EPI, int CPW, int CPWR>
__global__ __launch_bounds__(kMaxWaves * 64) void gemv_rider_kernel(const ua2_linear_args a, const int a_stride, const int red_off, const int rt,
                                                                    const ua2_linear_args ra, const int tile0, const int tile1, const int host_gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < host_gx) gemv_body<UA2_BF16, PRO, EPI, CPW, false>(a, smem, a_stride, red_off, rt, blockIdx.x, 0);
  else rider_body<CPWR>(ra, smem, tile0, tile1, (int)blockIdx.x - host_gx);
}

constexpr size_t kLdsABudget = 112 * 1024;   // activation tile budget (of 160 KiB; the rest holds the reduction buffers)

}  // namespace
int ua2_gemv_rows_per_tile(int dtype, int K) {
  const int kc = dtype == UA2_BF16 ? 32 : 16, bytes = dtype == UA2_BF16 ? 2 : 4;
  const size_t row_bytes = ((size_t)ua2_ceil_div(K, kc) * kc + 16 / bytes) * bytes;
  const int r = (int)(kLdsABudget / row_bytes);
  return r > 16 ? 16 : r;
}
// Rows up to which the launchers PREFER this kernel over the many-row ones (bf16; both give the same bits, so a pure cost choice).  The
// decode kernel stages every row of its tile in LDS in every workgroup: its frame grows ~0.14 ms per row (2.89 ms at 1 row, 3.58 at 5,
// 5.04 at 16), while the weights-stationary kernel runs 6 ... 16 rows in a flat 3.58-3.60 ms (tools/ubench/frame_vs_batch.py,
// profiles/r6_frame_vs_batch.txt) — until round 6 a 16-sequence batch decoded slower than a 64-sequence one.  UA2_GEMV_MAX_ROWS=n: sweeps.
int ua2_gemv_rows_preferred(int dtype, int K) {
  static const int pref = getenv("UA2_GEMV_MAX_ROWS") ? atoi(getenv("UA2_GEMV_MAX_ROWS")) : 5;       // read once
  const int r = ua2_gemv_rows_per_tile(dtype, K);
  return dtype == UA2_BF16 ? std::min(r, std::max(pref, 1)) : r;
}
namespace {
int rows_per_tile(int dtype, int K) { return ua2_gemv_rows_per_tile(dtype, K); }

struct Geometry {
  int waves, cpw;
};

// waves x cpw == nchunks when possible (no tail); prefer many waves for small grids (latency),
// fewer + deeper for large grids (two workgroups per CU overlap each other's prologue/epilogue).
Geometry pick_geometry(int nchunks, int blocks, int nt) {
  if (const char* e = getenv("UA2_GEMV_GEOM")) {  // experiment hook: "waves,cpw"
    int w = 0, c = 0;
    if (sscanf(e, "%d,%d", &w, &c) == 2 && w >= 1 && w <= kMaxWaves && (c == 4 || c == 8 || (c == 16 && nt == 1)) && w <= nchunks &&
        nchunks % w == 0 && (nchunks / w) % c == 0)
      return Geometry{w, c};
  }
  // measured (tools/ubench/gemv_shapes.py, profiles/r1_gemv_geometry.txt): grids larger than the CU
  // count stream best as 8-wave workgroups walking K in double-buffered rounds of 4 chunks (two
  // workgroups per CU overlap each other's reduce/epilogue); smaller grids as one wide burst.
  // (K = 2048 -> 64 chunks included: the local decoder's SwiGLU runs 13.1 us this way vs 16.0 us as one burst)
  if (blocks > 256 && nchunks >= 64 && nchunks % 32 == 0) return Geometry{8, 4};
  const int cap = (nt == 2) ? 8 : 16;
  const int cpws[3] = {4, 8, 16};
  Geometry best{0, 0};
  int best_score = -1;
  for (int w = 4; w <= kMaxWaves; ++w) {
    for (int ci = 0; ci < 3; ++ci) {
      const int c = cpws[ci];
      if (c > cap || w * c != nchunks) continue;
      int score = (blocks <= 320) ? w * 4 + c : (w >= 8 && w <= 12 ? 100 : 0) + c;
      if (score > best_score) { best_score = score; best = Geometry{w, c}; }
    }
  }
  if (best.waves) return best;
  // no exact tiling: 16 waves (or fewer if K is tiny), rounds of `cap` chunks with clamped loads
  int w = kMaxWaves;
  while (w > 4 && nchunks < w) w >>= 1;
  int per = (nchunks + w - 1) / w;
  int c = per <= 4 ? 4 : (per <= 8 ? 8 : cap);
  return Geometry{w, c};
}

template <int DT, int PRO, int EPI, int CPW, bool MR>
void launch_one(const ua2_linear_args& a, dim3 grid, int waves, int a_stride, int red_off, size_t smem, hipStream_t s, int rt) {
  constexpr auto kern = gemv_kernel<DT, PRO, EPI, CPW, MR>;
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, grid, dim3(waves * 64), smem, s, a, a_stride, red_off, rt);
  ua2_count_launch(UA2_CNT_GEMV);
}

template <int DT, int PRO, int EPI>
int launch_cpw(const ua2_linear_args& a, hipStream_t s) {
  constexpr int KC = Elem<DT>::KC, BYTES = Elem<DT>::BYTES;
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  const int nchunks = ua2_ceil_div(a.K, KC);
  const int ntiles = ua2_ceil_div(a.N, 16);
  const int gx = ntiles;
  // rows per workgroup and launch geometry depend on (N, K, dtype) only — never on M — so that a row
  // sees the same instruction stream and summation order whatever the batch around it
  const Geometry geo = pick_geometry(nchunks, gx, NT);
  // +16 B per row: breaks the power-of-two row stride (LDS bank conflicts across rows)
  const int a_stride = nchunks * KC + 16 / BYTES;
  const int rt = rows_per_tile(a.dtype, a.K);
  const int mtiles = ua2_ceil_div(a.M, rt);
  // LDS is sized for the rows actually present (occupancy: two single-row workgroups share a CU);
  // the size changes nothing a row computes
  const int red_off = (int)(((size_t)(a.M < rt ? a.M : rt) * a_stride * BYTES + 255) & ~(size_t)255);
  const size_t smem = (size_t)red_off + (size_t)(kMaxWaves * NT * 256 + 2 * kMaxWaves * 16 + 32) * sizeof(float);
  const dim3 grid(gx, mtiles);
  const bool mr = geo.waves <= 8 && geo.waves * geo.cpw < nchunks;
  if (mr) {
    if (geo.cpw == 4) launch_one<DT, PRO, EPI, 4, true>(a, grid, geo.waves, a_stride, red_off, smem, s, rt);
    else launch_one<DT, PRO, EPI, 8, true>(a, grid, geo.waves, a_stride, red_off, smem, s, rt);
  } else {
    switch (geo.cpw) {
      case 4: launch_one<DT, PRO, EPI, 4, false>(a, grid, geo.waves, a_stride, red_off, smem, s, rt); break;
      case 8: launch_one<DT, PRO, EPI, 8, false>(a, grid, geo.waves, a_stride, red_off, smem, s, rt); break;
      default:
        if constexpr (NT == 1) launch_one<DT, PRO, EPI, 16, false>(a, grid, geo.waves, a_stride, red_off, smem, s, rt);
        else launch_one<DT, PRO, EPI, 8, false>(a, grid, geo.waves, a_stride, red_off, smem, s, rt);
    }
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

template <int DT>
int launch_dt(const ua2_linear_args& a, hipStream_t s) {
  if (a.prologue == UA2_PRO_NORM) {
    if (a.epilogue == UA2_EPI_QKV_ROPE) return launch_cpw<DT, UA2_PRO_NORM, UA2_EPI_QKV_ROPE>(a, s);
    if (a.epilogue == UA2_EPI_SWIGLU) return launch_cpw<DT, UA2_PRO_NORM, UA2_EPI_SWIGLU>(a, s);
    if (a.epilogue == UA2_EPI_STORE) return launch_cpw<DT, UA2_PRO_NORM, UA2_EPI_STORE>(a, s);
    if (a.epilogue == UA2_EPI_GELU) return launch_cpw<DT, UA2_PRO_NORM, UA2_EPI_GELU>(a, s);
  } else if (a.prologue == UA2_PRO_CAST) {
    if (a.epilogue == UA2_EPI_RESIDUAL) return launch_cpw<DT, UA2_PRO_CAST, UA2_EPI_RESIDUAL>(a, s);
    if (a.epilogue == UA2_EPI_STORE) return launch_cpw<DT, UA2_PRO_CAST, UA2_EPI_STORE>(a, s);
    if (a.epilogue == UA2_EPI_SWIGLU) return launch_cpw<DT, UA2_PRO_CAST, UA2_EPI_SWIGLU>(a, s);     // codec GLU (no norm in front)
    if (a.epilogue == UA2_EPI_GELU) return launch_cpw<DT, UA2_PRO_CAST, UA2_EPI_GELU>(a, s);
    if (a.epilogue == UA2_EPI_QKV_ROPE) return launch_cpw<DT, UA2_PRO_CAST, UA2_EPI_QKV_ROPE>(a, s);
  } else if (a.prologue == UA2_PRO_LOCAL_ATTN) {
    if (a.epilogue == UA2_EPI_RESIDUAL) return launch_cpw<DT, UA2_PRO_LOCAL_ATTN, UA2_EPI_RESIDUAL>(a, s);
  } else if (a.prologue == UA2_PRO_SCALED) {
    if constexpr (DT == UA2_BF16) {
      if (a.epilogue == UA2_EPI_QKV_ROPE) return launch_cpw<DT, UA2_PRO_SCALED, UA2_EPI_QKV_ROPE>(a, s);
      if (a.epilogue == UA2_EPI_SWIGLU) return launch_cpw<DT, UA2_PRO_SCALED, UA2_EPI_SWIGLU>(a, s);
      if (a.epilogue == UA2_EPI_STORE) return launch_cpw<DT, UA2_PRO_SCALED, UA2_EPI_STORE>(a, s);
    }
  }
  return 1;  // combination not specialised here: the caller falls back to the general kernel
}

}  // namespace

// ---- riders (see gemv_rider_kernel) ------------------------------------------------------------------------------------------
namespace {
// host forms with a rider instantiation: (prologue, epilogue, chunks per wave) of a 16-wave single-burst launch.  Only the
// down-projection (K = 8192, 10.6 us: longer than a rider workgroup's own ~8 us) carries riders: on the 6-us launches (q|k|v and
// o-projection of the depth decoder: forms measured in round 6, profiles/r6_notes.md) a rider workgroup outlasts its host and the
// frame got SLOWER (3.18 against 3.11 ms).
int rider_host_form(const ua2_linear_args& a, const Geometry& geo) {
  if (geo.waves != 16) return -1;
  if (a.prologue == UA2_PRO_CAST && a.epilogue == UA2_EPI_RESIDUAL && geo.cpw == 16) return 0;
  return -1;
}
}  // namespace

// Can launches of `a`'s shape (the host) carry column tiles of `r` (the rider)?  A function of the two problems only, so a plan
// decides once.  Host: a bf16 launch of one row tile in a single-burst 16-wave geometry (rider_host_form) whose grid leaves CUs
// idle.  Rider: bf16 CAST / STORE (+ arg-max partials), no bias / hand-over, the 8-range x 12-chunk geometry its own launch would
// take (K = 3072), the same rows.
bool ua2_gemv_rider_ok(const ua2_linear_args& a, const ua2_linear_args& r) {
  static Ua2EnvInt off{"UA2_NO_RIDER", 0};
  if (off.set()) return false;
  if (a.dtype != UA2_BF16 || a.K % 32 || a.x_packed) return false;
  const int nch = a.K / 32, gx = ua2_ceil_div(a.N, 16);
  const Geometry geo = pick_geometry(nch, gx, 1);
  if (geo.waves * geo.cpw != nch || gx >= 256 || rider_host_form(a, geo) < 0) return false;
  if (a.M > ua2_gemv_rows_preferred(a.dtype, a.K) || r.M != a.M) return false;      // the host launch must be one the launchers give this kernel
  if (r.dtype != UA2_BF16 || r.prologue != UA2_PRO_CAST || r.epilogue != UA2_EPI_STORE || r.K != 3072 || !r.x || r.ldx % 4) return false;
  if (r.bias || r.y_norm_w || r.x_packed || r.M > rows_per_tile(r.dtype, r.K)) return false;
  const Geometry rg = pick_geometry(r.K / 32, ua2_ceil_div(r.N, 16), 1);
  return rg.waves == 8 && rg.waves * 12 == r.K / 32;       // what the rider's own launch sums like (8 ranges of 12 chunks)
}

// The host launch with rider tiles [tile0, tile1) of `r` on workgroups past its grid.  0 = launched, 1 = not applicable.
int ua2_gemv_launch_with_rider(const ua2_linear_args& a, const ua2_linear_args& r, int tile0, int tile1, hipStream_t s) {
  if (!ua2_gemv_rider_ok(a, r)) return 1;
  const int nchunks = a.K / 32, gx = ua2_ceil_div(a.N, 16);
  const int form = rider_host_form(a, pick_geometry(nchunks, gx, 1));
  const int a_stride = nchunks * 32 + 8;
  const int red_off = (int)(((size_t)a.M * a_stride * 2 + 255) & ~(size_t)255);
  const size_t host_smem = (size_t)red_off + (size_t)(kMaxWaves * 256 + 2 * kMaxWaves * 16 + 32) * sizeof(float);
  const size_t rider_smem = (((size_t)r.M * (r.K + 8) * 2 + 255) & ~(size_t)255) + (size_t)2 * 8 * 256 * sizeof(float);
  const size_t smem = std::max(host_smem, rider_smem);
  const int riders = tile1 > tile0 ? ua2_ceil_div(tile1 - tile0, 2) : 0;
  const int rt = rows_per_tile(a.dtype, a.K);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(gx + riders, 1), dim3(kMaxWaves * 64), smem, s, a, a_stride, red_off, rt, r, tile0, tile1, gx);
  };
#define UA2_RIDER(PRO_, EPI_, CPW_) { constexpr auto kern = gemv_rider_kernel<PRO_, EPI_, CPW_, 12>; ua2_allow_big_lds<kern>(); go(kern); }
  if (form != 0) return 1;
  UA2_RIDER(UA2_PRO_CAST, UA2_EPI_RESIDUAL, 16)
#undef UA2_RIDER
  ua2_count_launch(UA2_CNT_GEMV);
  UA2_LAUNCH_CHECK();
  return 0;
}

ua2_gemv_geometry ua2_pick_gemv_geometry(int dtype, int N, int K, int nt) {
  const int kc = dtype == UA2_BF16 ? 32 : 16;
  const Geometry g = pick_geometry(ua2_ceil_div(K, kc), ua2_ceil_div(N, 16), nt);
  return ua2_gemv_geometry{g.waves, g.cpw};
}

// Returns 0 if launched, 1 if this problem is outside the decode regime (caller uses the general
// kernel), negative on error.
int ua2_gemv_try_launch(const ua2_linear_args& a, hipStream_t s) {
  if (rows_per_tile(a.dtype, a.K) < 1) return 1;   // a single row does not fit the LDS budget
  if (a.dtype == UA2_BF16) return launch_dt<UA2_BF16>(a, s);
  if (a.dtype == UA2_F32) return launch_dt<UA2_F32>(a, s);
  return 1;
}
