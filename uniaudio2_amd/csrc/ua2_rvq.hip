// Residual-VQ nearest-codeword search and lookup.
//
// Replaces, for both RVQ flavours of the codec:
//   tools/tokenizer/MimiCodec/model/quantization/core_vq.py:179-185 (cdist + argmin), :365-376 (residual
//   loop), :378-384 / :198-206 (lookup + sum), and the un-vendored vector_quantize_pytorch.ResidualVQ the
//   live codec calls at ReasoningCodec_film/models/AudioDiffusion1D.py:388,529,535,544 (quantise) and
//   :577-583 (get_output_from_indices) — the projections around it are plain Linears and stay outside.
//
// Arithmetic contract = oracle/rvq_oracle.c, bit for bit: d2 = sum_k fma(x_k - e_k, x_k - e_k, acc),
// k ascending; arg-min with the lowest index on ties; residual -= e, quantised += e per level.
//
// MI355X mapping: the search is an HBM/L2-bound scan of a small codebook (8192 x 32 fp32 = 1 MiB per
// level) — not a GEMM: expanding |x-e|^2 into x.e products to reach MFMA would change the rounding
// and lose bit-exactness.  A workgroup owns VB vectors (8, 2 or 1: the launcher picks the largest that still gives
// >= 256 workgroups — one 10-s clip is only 125 vectors and ran on 16 CUs with VB = 8; residuals in LDS); each lane owns codewords
// c, c+256, ... and walks k with coalesced loads from the k-major codebook copy ([L][D][C]: 64 lanes
// read 64 consecutive codewords of one k), reusing every loaded value for the VB vectors; the
// arg-min is a wavefront shuffle reduction on (distance, index) pairs, then one LDS hop across the
// four waves.
//
// Few vectors (one 10-s clip = 125 vectors): the form above gives 16-125 workgroups, each scanning all 6 MiB of
// codebooks with 32 serial codeword iterations per lane — latency-bound at 1.5 % of the vector peak (255 us, unchanged over
// two rounds).  `rvq_encode_split_kernel` splits the CODEBOOK over S workgroups per group of 8 vectors (grid = groups x S
// ~ 256: one per CU): every workgroup scans C / S codewords of the level for its 8 vectors (each loaded value reused 8
// times), publishes its (distance, index) candidates with one 64-bit agent-scope atomic min per vector — the key is
// distance bits << 32 | index, so the minimum is the smallest distance and, among equal distances, the lowest index: the
// oracle's tie rule — and the S workgroups of a group meet at a counter (arrivals after `s_waitcnt vmcnt(0)`, relaxed
// polls: the cheap forms of profiles/r2_gridbar.txt) before each of them applies the same residual update.  The per-codeword
// arithmetic is untouched: bit-exact against oracle/rvq_oracle.c.  The S x groups workgroups are co-resident by
// construction (<= 512 workgroups of 256 threads), so the spin cannot deadlock; it is bounded anyway (a stuck peer
// flags an error code instead of hanging the GPU).
#include <stdlib.h>

#include "ua2_common.h"

namespace {

constexpr int kThreads = 256;

// number of split-codebook launches whose bounded spin expired and that were redone by the fall-through launch (ua2_rvq_fallbacks)
__device__ unsigned g_rvq_fallbacks = 0;

__device__ __forceinline__ void argmin_pair(float& v, int& i, float ov, int oi) {
  if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <int kVB>
__global__ __launch_bounds__(kThreads) void rvq_encode_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                              const float* __restrict__ embT, int64_t N, int L, int C,
                                                              int D, int32_t* __restrict__ codes,
                                                              float* __restrict__ quantized, const unsigned* gate) {
  // `gate` (optional): the error flag of a split-codebook launch issued just before on the same stream.  Clean (0xffffffff):
  // nothing to do.  Tripped: the split launch gave up on a peer (bounded spin) and its codes are not to be trusted — this
  // launch recomputes every vector with the self-contained scan, so the caller never sees rc 0 with wrong codes.
  if (gate) {
    if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xffffffffu) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_rvq_fallbacks, 1u);
  }
  extern __shared__ float sm[];
  float* res = sm;                 // [kVB][D]
  float* qs = res + kVB * D;       // [kVB][D]
  float* wv = qs + kVB * D;        // [4][kVB] wave minima
  int* wi = reinterpret_cast<int*>(wv + 4 * kVB);
  int* best_i = wi + 4 * kVB;      // [kVB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * kVB;
  const int nv = (int)min((int64_t)kVB, N - n0);
  for (int idx = tid; idx < kVB * D; idx += kThreads) {
    const int v = idx / D, k = idx - v * D;
    res[idx] = (v < nv) ? x[(n0 + v) * D + k] : 0.f;
    qs[idx] = 0.f;
  }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    const float* eT = embT + (size_t)l * D * C;
    float bv[kVB];
    int bi[kVB];
#pragma unroll
    for (int v = 0; v < kVB; ++v) { bv[v] = INFINITY; bi[v] = 0x7fffffff; }
    for (int c = tid; c < C; c += kThreads) {
      float acc[kVB];
#pragma unroll
      for (int v = 0; v < kVB; ++v) acc[v] = 0.f;
#pragma unroll 32
      for (int k = 0; k < D; ++k) {
        const float e = eT[(size_t)k * C + c];
#pragma unroll
        for (int v = 0; v < kVB; ++v) {
          const float d = __fsub_rn(res[v * D + k], e);
          acc[v] = __fmaf_rn(d, d, acc[v]);
        }
      }
#pragma unroll
      for (int v = 0; v < kVB; ++v)
        if (acc[v] < bv[v]) { bv[v] = acc[v]; bi[v] = c; }   // c ascends per lane: strict '<' keeps the first
    }
#pragma unroll
    for (int v = 0; v < kVB; ++v) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) argmin_pair(bv[v], bi[v], __shfl_xor(bv[v], o), __shfl_xor(bi[v], o));
      if (lane == 0) { wv[wave * kVB + v] = bv[v]; wi[wave * kVB + v] = bi[v]; }
    }
    __syncthreads();
    if (tid < kVB) {
      float v0 = wv[tid];
      int i0 = wi[tid];
      for (int w = 1; w < 4; ++w) argmin_pair(v0, i0, wv[w * kVB + tid], wi[w * kVB + tid]);
      best_i[tid] = i0;
      if (tid < nv) codes[(n0 + tid) * L + l] = i0;
    }
    __syncthreads();
    const float* eR = emb + (size_t)l * C * D;
    for (int idx = tid; idx < kVB * D; idx += kThreads) {
      const int v = idx / D, k = idx - v * D;
      const float e = eR[(size_t)best_i[v] * D + k];
      res[idx] = __fsub_rn(res[idx], e);      // residual = residual - quantized   (core_vq.py:372)
      qs[idx] = __fadd_rn(qs[idx], e);
    }
    __syncthreads();
  }
  if (quantized)
    for (int idx = tid; idx < nv * D; idx += kThreads) quantized[n0 * D + idx] = qs[idx];
}

constexpr int kSplitVB = 8;
constexpr unsigned long long kKeyInit = ~0ull;          // the workspace is memset to 0xff: keys start at "+inf, no index"

// workspace layout: keys [L][groups][32][8] u64 (this kernel uses [l][grp][0][v] as the group's atomic-min cells; the
// one-codeword-per-thread kernel below uses a slot per split), then counters [L][groups] u32, then one u32 error flag
// (0xffffffff = clean); all bytes 0xff before the launch
template <int kVB>
__global__ __launch_bounds__(kThreads) void rvq_encode_split_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                                    const float* __restrict__ embT, int64_t N, int L, int C, int D,
                                                                    int32_t* __restrict__ codes, float* __restrict__ quantized,
                                                                    unsigned long long* keys, unsigned* counters, unsigned* err,
                                                                    const int spin_limit) {
  extern __shared__ float sm[];
  float* res = sm;                 // [kVB][D]
  float* qs = res + kVB * D;       // [kVB][D]
  float* wv = qs + kVB * D;        // [4][kVB] wave minima
  int* wi = reinterpret_cast<int*>(wv + 4 * kVB);
  int* best_i = wi + 4 * kVB;      // [kVB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, S = gridDim.y, cs = blockIdx.y, groups = gridDim.x;
  const int64_t n0 = (int64_t)grp * kVB;
  const int nv = (int)min((int64_t)kVB, N - n0);
  const int c_lo = (int)((int64_t)C * cs / S), c_hi = (int)((int64_t)C * (cs + 1) / S);
  if (spin_limit <= 0 && tid == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // test hook: a bound of 0 = "gave up"
  for (int idx = tid; idx < kVB * D; idx += kThreads) {
    const int v = idx / D, k = idx - v * D;
    res[idx] = (v < nv) ? x[(n0 + v) * D + k] : 0.f;
    qs[idx] = 0.f;
  }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    const float* eT = embT + (size_t)l * D * C;
    float bv[kVB];
    int bi[kVB];
#pragma unroll
    for (int v = 0; v < kVB; ++v) { bv[v] = INFINITY; bi[v] = 0x7fffffff; }
    for (int c = c_lo + tid; c < c_hi; c += kThreads) {
      float acc[kVB];
#pragma unroll
      for (int v = 0; v < kVB; ++v) acc[v] = 0.f;
#pragma unroll 32
      for (int k = 0; k < D; ++k) {
        const float e = eT[(size_t)k * C + c];
#pragma unroll
        for (int v = 0; v < kVB; ++v) {
          const float d = __fsub_rn(res[v * D + k], e);
          acc[v] = __fmaf_rn(d, d, acc[v]);
        }
      }
#pragma unroll
      for (int v = 0; v < kVB; ++v)
        if (acc[v] < bv[v]) { bv[v] = acc[v]; bi[v] = c; }   // c ascends per lane: strict '<' keeps the first
    }
#pragma unroll
    for (int v = 0; v < kVB; ++v) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) argmin_pair(bv[v], bi[v], __shfl_xor(bv[v], o), __shfl_xor(bi[v], o));
      if (lane == 0) { wv[wave * kVB + v] = bv[v]; wi[wave * kVB + v] = bi[v]; }
    }
    __syncthreads();
    unsigned long long* lkeys = keys + (((size_t)l * groups + grp) * 32) * kVB;
    unsigned* cnt = counters + (size_t)l * groups + grp;
    if (tid < kVB) {
      float v0 = wv[tid];
      int i0 = wi[tid];
      for (int w = 1; w < 4; ++w) argmin_pair(v0, i0, wv[w * kVB + tid], wi[w * kVB + tid]);
      // distances are >= +0: their bit patterns order like the floats; NaN (never the minimum in the oracle either) sorts last
      const unsigned long long key = ((unsigned long long)__float_as_uint(v0) << 32) | (unsigned)i0;
      if (i0 != 0x7fffffff) __hip_atomic_fetch_min(lkeys + tid, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the candidates are performed before this workgroup arrives
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // 0xffffffff + S arrivals = S - 1
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(S - 1)) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > spin_limit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    if (tid < kVB) {
      const unsigned long long key = __hip_atomic_load(lkeys + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int i0 = (int)(unsigned)(key & 0xffffffffull);
      best_i[tid] = (key == kKeyInit) ? 0 : i0;             // all-NaN column: the oracle's `bi = 0`
      if (cs == 0 && tid < nv) codes[(n0 + tid) * L + l] = best_i[tid];
    }
    __syncthreads();
    const float* eR = emb + (size_t)l * C * D;
    for (int idx = tid; idx < kVB * D; idx += kThreads) {
      const int v = idx / D, k = idx - v * D;
      const float e = eR[(size_t)best_i[v] * D + k];
      res[idx] = __fsub_rn(res[idx], e);      // residual = residual - quantized   (core_vq.py:372)
      qs[idx] = __fadd_rn(qs[idx], e);
    }
    __syncthreads();
  }
  if (quantized && cs == 0)
    for (int idx = tid; idx < nv * D; idx += kThreads) quantized[n0 * D + idx] = qs[idx];
}

// One codeword per thread (S = C / 256 splits): the thread's codeword lives in registers, and the NEXT level's codeword is
// requested while this level's candidates travel through the atomics — the codebook loads do not depend on the residual,
// only the arithmetic does.  What the generic split kernel above spends per level is not the scan (0.9 us of VALU) but the
// dependent load batches in front of it (k walked 8 loads at a time: four memory round trips per codeword): 25-30 us per
// level, measured (profiles/r3_notes.md).
template <int kVB, int kD>
__global__ __launch_bounds__(kThreads, (kD <= 32 ? 4 : 2)) void rvq_encode_split1_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                                     const float* __restrict__ embT, int64_t N, int L, int C,
                                                                     int32_t* __restrict__ codes, float* __restrict__ quantized,
                                                                     unsigned long long* keys, unsigned* counters, unsigned* err,
                                                                     const int spin_limit) {
  __shared__ float res[kVB * kD], qs[kVB * kD], dist[kVB * kThreads];
  __shared__ int best_i[kVB];
  __shared__ unsigned long long skeys[kThreads];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, S = gridDim.y, cs = blockIdx.y, groups = gridDim.x;
  const int64_t n0 = (int64_t)grp * kVB;
  const int nv = (int)min((int64_t)kVB, N - n0);
  const int c = cs * kThreads + tid;                       // this thread's codeword (C == S * 256)
  if (spin_limit <= 0 && tid == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // test hook: a bound of 0 = "gave up"
  float e_cur[kD], e_next[kD];
#pragma unroll
  for (int k = 0; k < kD; ++k) e_cur[k] = embT[(size_t)k * C + c];      // level 0, all k in flight at once
  for (int idx = tid; idx < kVB * kD; idx += kThreads) {
    const int v = idx / kD, k = idx - v * kD;
    res[idx] = (v < nv) ? x[(n0 + v) * kD + k] : 0.f;
    qs[idx] = 0.f;
  }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    // Distances vector by vector into LDS (not unrolled: fully unrolled the compiler keeps all kVB x kD residual values in
    // registers — 256 VGPRs, one workgroup per CU, and the co-residency the exchange relies on is gone; the residual row is a
    // broadcast LDS read).  The arg-min then runs two vectors per wave with the shuffle chains of the two interleaved — done
    // per vector inside the loop above it was 8 x 6 dependent ds_bpermute pairs, ~2.4 us per level.
#pragma unroll 1
    for (int v = 0; v < kVB; ++v) {
      float a0 = 0.f;
#pragma unroll
      for (int k = 0; k < kD; ++k) {
        const float d = __fsub_rn(res[v * kD + k], e_cur[k]);
        a0 = __fmaf_rn(d, d, a0);
      }
      dist[v * kThreads + tid] = a0;
    }
    __syncthreads();
    // Exchange inside the group: every workgroup publishes its kVB candidates as 64-bit keys (distance bits << 32 | index: a
    // smaller key is a smaller distance and, at equal distance, the lower index — the oracle's tie rule; NaN bit patterns sort
    // last, as a NaN never wins there) with write-through stores into its own slots, and every workgroup reads all S x kVB
    // slots — one per thread — polling until the slot differs from the 0xff.. the workspace was memset to.  A valid key never
    // equals it (index < 2^31).  Two memory round trips per level (store -> visible, poll) instead of the five of an
    // atomic-min + arrival-counter + key read-back chain (measured 14.6 us per level that way).
    unsigned long long* lkeys = keys + (((size_t)l * groups + grp) * S) * kVB;        // [S][kVB]
    {
      static_assert(kVB == 8, "two vectors per wave, four waves");
      unsigned long long k2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* dv = dist + (2 * wave + h) * kThreads;
        unsigned long long best = kKeyInit;
#pragma unroll
        for (int j = 0; j < kThreads / 64; ++j) {
          const int t = lane + 64 * j;
          best = min(best, ((unsigned long long)__float_as_uint(dv[t]) << 32) | (unsigned)(cs * kThreads + t));
        }
        k2[h] = best;
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) k2[h] = min(k2[h], (unsigned long long)__shfl_xor((long long)k2[h], o));
      }
      if (lane < 2) __hip_atomic_store(lkeys + (size_t)cs * kVB + 2 * wave + lane, lane ? k2[1] : k2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the next level's codeword goes out now: its round trip runs under the exchange and the residual update below
    const float* eT = embT + (size_t)min(l + 1, L - 1) * kD * C;
#pragma unroll
    for (int k = 0; k < kD; ++k) e_next[k] = eT[(size_t)k * C + c];
    unsigned long long mine = kKeyInit;
    if (tid < S * kVB) {
      int spins = 0;
      while ((mine = __hip_atomic_load(lkeys + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kKeyInit) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    skeys[tid] = mine;
    __syncthreads();
    if (tid < kVB) {
      unsigned long long best = kKeyInit;
      for (int sp = 0; sp < S; ++sp) best = min(best, skeys[sp * kVB + tid]);
      best_i[tid] = (best == kKeyInit) ? 0 : (int)(unsigned)(best & 0xffffffffull);
      if (cs == 0 && tid < nv) codes[(n0 + tid) * L + l] = best_i[tid];
    }
    __syncthreads();
    const float* eR = emb + (size_t)l * C * kD;
    for (int idx = tid; idx < kVB * kD; idx += kThreads) {
      const int v = idx / kD, k = idx - v * kD;
      const float e = eR[(size_t)best_i[v] * kD + k];
      res[idx] = __fsub_rn(res[idx], e);      // residual = residual - quantized   (core_vq.py:372)
      qs[idx] = __fadd_rn(qs[idx], e);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kD; ++k) e_cur[k] = e_next[k];
  }
  if (quantized && cs == 0)
    for (int idx = tid; idx < nv * kD; idx += kThreads) quantized[n0 * kD + idx] = qs[idx];
}

__global__ void rvq_decode_kernel(const int32_t* __restrict__ codes, const float* __restrict__ emb, int64_t N, int L,
                                  int C, int D, float* __restrict__ out) {
  const int64_t total = N * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = idx / D;
    const int k = (int)(idx - n * D);
    float q = 0.f;
    for (int l = 0; l < L; ++l) q = __fadd_rn(q, emb[((size_t)l * C + codes[n * L + l]) * D + k]);  // core_vq.py:380-383
    out[idx] = q;
  }
}

}  // namespace

// groups of 8 vectors, codebook splits (a power of two), for the split form; S == 1: not worth it / not possible
static void rvq_split_plan(int64_t N, int C, int* groups, int* S) {
  *groups = (int)((N + kSplitVB - 1) / kSplitVB);
  int s = 1;
  while (s < 32 && (int64_t)*groups * (s * 2) <= 384 && C / (s * 2) >= kThreads) s *= 2;
  *S = (*groups < 128) ? s : 1;
}

extern "C" size_t ua2_rvq_workspace_bytes(int64_t N, int32_t L) {
  if (N <= 0 || L <= 0) return 0;
  const size_t groups = (size_t)((N + kSplitVB - 1) / kSplitVB);
  return (size_t)L * groups * 32 * kSplitVB * 8 + (size_t)L * groups * 4 + 8;     // keys [L][groups][<= 32 splits][8], counters, error flag
}

extern "C" int ua2_rvq_encode(const float* x, const float* emb, const float* embT, int64_t N, int32_t L, int32_t C,
                              int32_t D, int32_t* codes, float* quantized, void* workspace, size_t workspace_bytes, void* stream) {
  UA2_CHECK(x && emb && embT && codes && N > 0 && L > 0 && C > 0 && D > 0 && D <= 1024, "ua2_rvq_encode: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // the self-contained scan; with `gate` = a split launch's error flag it is that launch's fall-through (a no-op when clean)
  auto launch_plain = [&](const unsigned* gate) -> int {
    int vb = 8;
    while (vb > 1 && (N + vb - 1) / vb < 256) vb = vb == 8 ? 2 : 1;
    const size_t smem = (size_t)(2 * vb * D + 4 * vb) * sizeof(float) + (size_t)(4 * vb + vb) * sizeof(int);
    UA2_CHECK(smem <= 64 * 1024, "ua2_rvq_encode: D=%d too large", D);
    const int blocks = (int)((N + vb - 1) / vb);
    if (vb == 8) hipLaunchKernelGGL(rvq_encode_kernel<8>, dim3(blocks), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes, quantized, gate);
    else if (vb == 2) hipLaunchKernelGGL(rvq_encode_kernel<2>, dim3(blocks), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes, quantized, gate);
    else hipLaunchKernelGGL(rvq_encode_kernel<1>, dim3(blocks), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes, quantized, gate);
    UA2_LAUNCH_CHECK();
    return 0;
  };
  int groups = 0, S = 1;
  rvq_split_plan(N, C, &groups, &S);
  // bounded spins of the split kernels; UA2_RVQ_SPIN_LIMIT (read per call) lets a test force the time-out path
  int spin1 = 1 << 20, spin2 = 1 << 22;
  if (const char* e = getenv("UA2_RVQ_SPIN_LIMIT")) spin1 = spin2 = std::max(0, atoi(e));
  static const bool no_split = getenv("UA2_RVQ_NO_SPLIT") != nullptr;      // A/B hook
  if (S > 1 && workspace && !no_split) {
    const size_t need = ua2_rvq_workspace_bytes(N, L);
    UA2_CHECK(workspace_bytes >= need, "ua2_rvq_encode: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    UA2_HIP(hipMemsetAsync(workspace, 0xff, need, s));
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
    unsigned* counters = reinterpret_cast<unsigned*>(keys + (size_t)L * groups * 32 * kSplitVB);
    unsigned* err = counters + (size_t)L * groups;
    const int s1 = C / kThreads;                            // one codeword per thread
    // every workgroup of the grid must be resident at once (they wait for each other): cap the grid by what the device holds
    static int cus = 0, occ32 = 0, occ64 = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      int dev = 0;
      UA2_HIP(hipGetDevice(&dev));
      UA2_HIP(hipGetDeviceProperties(&prop, dev));
      UA2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ32, rvq_encode_split1_kernel<kSplitVB, 32>, kThreads, 0));
      UA2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ64, rvq_encode_split1_kernel<kSplitVB, 64>, kThreads, 0));
      cus = prop.multiProcessorCount;
    }
    const int64_t room1 = (int64_t)cus * (D == 32 ? occ32 : occ64) / 2;      // half of the device: other streams may hold the rest
    if ((D == 32 || D == 64) && C % kThreads == 0 && s1 * kSplitVB <= kThreads && (int64_t)groups * s1 <= room1) {
      if (D == 32) hipLaunchKernelGGL((rvq_encode_split1_kernel<kSplitVB, 32>), dim3(groups, s1), dim3(kThreads), 0, s, x, emb, embT, N, L, C,
                                      codes, quantized, keys, counters, err, spin1);
      else hipLaunchKernelGGL((rvq_encode_split1_kernel<kSplitVB, 64>), dim3(groups, s1), dim3(kThreads), 0, s, x, emb, embT, N, L, C, codes,
                              quantized, keys, counters, err, spin1);
      UA2_LAUNCH_CHECK();
      return launch_plain(err);   // fall-through: redoes everything if a spin expired, returns at once otherwise
    }
    const size_t smem = (size_t)(2 * kSplitVB * D + 4 * kSplitVB) * sizeof(float) + (size_t)(4 * kSplitVB + kSplitVB) * sizeof(int);
    UA2_CHECK(smem <= 64 * 1024, "ua2_rvq_encode: D=%d too large", D);
    int occg = 0;
    UA2_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occg, rvq_encode_split_kernel<kSplitVB>, kThreads, smem));
    while (S > 1 && (int64_t)groups * S > (int64_t)cus * occg / 2) S /= 2;
    hipLaunchKernelGGL(rvq_encode_split_kernel<kSplitVB>, dim3(groups, S), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes,
                       quantized, keys, counters, err, spin2);
    UA2_LAUNCH_CHECK();
    return launch_plain(err);
  }
  return launch_plain(nullptr);
}

extern "C" int ua2_rvq_fallbacks(uint32_t* out) {
  UA2_CHECK(out != nullptr, "ua2_rvq_fallbacks: NULL argument");
  UA2_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rvq_fallbacks), sizeof(unsigned)));   // synchronous: a test / health-check call
  return 0;
}

extern "C" int ua2_rvq_decode(const int32_t* codes, const float* emb, int64_t N, int32_t L, int32_t C, int32_t D,
                              float* out, void* stream) {
  UA2_CHECK(codes && emb && out && N > 0 && L > 0 && C > 0 && D > 0, "ua2_rvq_decode: bad arguments");
  const int64_t total = N * D;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(rvq_decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, codes, emb, N, L, C, D, out);
  UA2_LAUNCH_CHECK();
  return 0;
}
