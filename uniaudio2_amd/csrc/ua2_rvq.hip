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
#include "ua2_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void argmin_pair(float& v, int& i, float ov, int oi) {
  if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <int kVB>
__global__ __launch_bounds__(kThreads) void rvq_encode_kernel(const float* __restrict__ x, const float* __restrict__ emb,
                                                              const float* __restrict__ embT, int64_t N, int L, int C,
                                                              int D, int32_t* __restrict__ codes,
                                                              float* __restrict__ quantized) {
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
#pragma unroll 8
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

extern "C" int ua2_rvq_encode(const float* x, const float* emb, const float* embT, int64_t N, int32_t L, int32_t C,
                              int32_t D, int32_t* codes, float* quantized, void* stream) {
  UA2_CHECK(x && emb && embT && codes && N > 0 && L > 0 && C > 0 && D > 0 && D <= 1024, "ua2_rvq_encode: bad arguments");
  int vb = 8;
  while (vb > 1 && (N + vb - 1) / vb < 256) vb = vb == 8 ? 2 : 1;
  const size_t smem = (size_t)(2 * vb * D + 4 * vb) * sizeof(float) + (size_t)(4 * vb + vb) * sizeof(int);
  UA2_CHECK(smem <= 64 * 1024, "ua2_rvq_encode: D=%d too large", D);
  const int blocks = (int)((N + vb - 1) / vb);
  hipStream_t s = (hipStream_t)stream;
  if (vb == 8) hipLaunchKernelGGL(rvq_encode_kernel<8>, dim3(blocks), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes, quantized);
  else if (vb == 2) hipLaunchKernelGGL(rvq_encode_kernel<2>, dim3(blocks), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes, quantized);
  else hipLaunchKernelGGL(rvq_encode_kernel<1>, dim3(blocks), dim3(kThreads), smem, s, x, emb, embT, N, L, C, D, codes, quantized);
  UA2_LAUNCH_CHECK();
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
