// Top-k sampling tail of a frame (model_new.py:146-187 sample_topk / audio_sample_topk with topk > 1):
//   logits / T  ->  [:forbid_prefix] = -inf  ->  keep logits >= k-th largest (ties at the threshold are
//   all kept, `logits < topk_value` is what gets removed, :150,182)  ->  softmax  ->  arg-max(probs / q),
//   q ~ Exp(1) (:141-143, the "exponential race" = one multinomial draw)  ->  int32 id,
// fused with the next-step embedding gather (:640,662-663) like ua2_argmax_embed.
//
// The k-th largest value is found exactly with a 4-pass radix select (8 bits per pass) on the
// order-preserving integer image of the fp32 logits, one workgroup per row (the row is L2-resident:
// 513 KB for the text head); softmax normalisation is dropped (it does not move the arg-max).
// Randomness: counter-based Philox4x32-10 keyed by (seed + the 64-bit device word counter[1..2], draw index
// counter[0] from the device frame counter, row, stream) — reproducible under hipGraph replay and independent of launch geometry.  It cannot
// reproduce torch's generator stream; parity with the reference is distributional (tests).
#include "ua2_common.h"

namespace {

__device__ __forceinline__ unsigned key_of(float f) {   // monotone: larger float -> larger key
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                           unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <int DT>
__global__ __launch_bounds__(1024) void sample_topk_kernel(const float* __restrict__ logits, int ld, int V, int topk,
                                                           float temperature, const int32_t* __restrict__ forbid,
                                                           unsigned long long seed, const int32_t* __restrict__ counter,
                                                           int stream_id, int32_t* __restrict__ out_tokens, int out_ld,
                                                           int out_col, const void* __restrict__ emb, int emb_off, int C,
                                                           float* __restrict__ next_h, int row_key_shift) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_remaining;
  __shared__ float red_v[16];
  __shared__ int red_i[16];
  __shared__ int tok_s;
  const int m = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* row = logits + (size_t)m * ld;
  const int fb = forbid ? forbid[m] : 0;
  // ---- exact k-th largest key among columns [fb, V) -------------------------------------------
  if (tid == 0) { sel_prefix = 0u; sel_remaining = (unsigned)topk; }
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = sel_prefix, shift = pass * 8;
    const unsigned mask_hi = (pass == 3) ? 0u : (0xffffffffu << (shift + 8));
    for (int c = fb + tid; c < V; c += nt) {
      const unsigned k = key_of(row[c] / temperature);
      if ((k & mask_hi) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned rem = sel_remaining;
      int b = 255;
      for (; b > 0; --b) {                       // walk buckets from the largest digit down
        if (hist[b] >= rem) break;
        rem -= hist[b];
      }
      sel_prefix = prefix | ((unsigned)b << shift);
      sel_remaining = rem;
    }
    __syncthreads();
  }
  const unsigned kth = sel_prefix;               // key of the k-th largest scaled logit
  // ---- exponential race among the kept columns ------------------------------------------------
  float mx = -INFINITY;
  for (int c = fb + tid; c < V; c += nt) mx = fmaxf(mx, row[c] / temperature);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red_v[tid >> 6] = mx;
  __syncthreads();
  mx = red_v[0];
  for (int w = 1; w < (nt >> 6); ++w) mx = fmaxf(mx, red_v[w]);
  __syncthreads();
  const unsigned draw = (unsigned)counter[0];
  // device-resident part of the key (ua2_stage3_set_sampling writes it): added to the by-value seed
  seed += (unsigned long long)(unsigned)counter[1] | ((unsigned long long)(unsigned)counter[2] << 32);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = fb + tid; c < V; c += nt) {
    const float l = row[c] / temperature;
    if (key_of(l) < kth) continue;               // removed: strictly below the k-th value
    unsigned r[4];
    philox4x32((unsigned)c, draw, (unsigned)(m >> row_key_shift), (unsigned)stream_id, (unsigned)seed, (unsigned)(seed >> 32), r);
    const float u = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);    // (0, 1)
    const float q = -logf(u);                                                // Exp(1)
    const float score = expf(l - mx) / q;                                    // probs / q up to the softmax constant
    if (score > bv || (score == bv && c < bi)) { bv = score; bi = c; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < (nt >> 6); ++w)
      if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
    tok_s = bi;
    out_tokens[(size_t)m * out_ld + out_col] = bi;
  }
  __syncthreads();
  if (emb) {
    const size_t er = (size_t)tok_s + (size_t)emb_off;
    for (int c = tid; c < C; c += nt) next_h[(size_t)m * C + c] = load_elem<DT>(emb, er * C + c);
  }
}

}  // namespace

extern "C" int ua2_sample_topk(int dtype, int32_t M, const float* logits, int32_t ld, int32_t V, int32_t topk,
                               float temperature, const int32_t* forbid, uint64_t seed, const int32_t* counter,
                               int32_t stream_id, int32_t* out_tokens, int32_t out_ld, int32_t out_col, const void* emb,
                               int32_t emb_row_offset, int32_t C, float* next_h, int32_t row_key_shift, void* stream) {
  UA2_CHECK(M > 0 && logits && V > 0 && out_tokens && counter && row_key_shift >= 0 && row_key_shift <= 1, "ua2_sample_topk: bad arguments");
  UA2_CHECK(temperature > 0.f, "temperature must be > 0");                       // model_new.py:165-166
  UA2_CHECK(topk >= 1 && topk <= V, "topk must be in 1..%d", V);                 // :177-178 (per-row forbid checked by the host)
  UA2_CHECK(!emb || next_h, "ua2_sample_topk: next_h is NULL");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UA2_BF16)
    hipLaunchKernelGGL((sample_topk_kernel<UA2_BF16>), dim3(M), dim3(1024), 0, s, logits, ld, V, topk, temperature, forbid,
                       (unsigned long long)seed, counter, stream_id, out_tokens, out_ld, out_col, emb, emb_row_offset, C, next_h, row_key_shift);
  else if (dtype == UA2_F32)
    hipLaunchKernelGGL((sample_topk_kernel<UA2_F32>), dim3(M), dim3(1024), 0, s, logits, ld, V, topk, temperature, forbid,
                       (unsigned long long)seed, counter, stream_id, out_tokens, out_ld, out_col, emb, emb_row_offset, C, next_h, row_key_shift);
  else {
    ua2_set_error("ua2_sample_topk: bad dtype %d", dtype);
    return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}
