// Small row-wise kernels of the decode frame: embedding merge, final-norm + step-mask blends,
// greedy sampling tail + next-step embedding gather.  All are a few KB of traffic; what matters
// is that each replaces a chain of 5-10 tiny PyTorch launches in the reference with one.
#include <stdarg.h>
#include <string.h>

#include <algorithm>

#include "ua2_common.h"

static thread_local char g_err[512] = "";

void ua2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ua2_last_error(void) { return g_err; }
extern "C" int ua2_version(void) { return UA2_VERSION; }

std::atomic<int64_t> g_ua2_launches[UA2_CNT_N];
std::atomic<int> g_ua2_env_gen{0};
extern "C" int64_t ua2_debug_kernel_launches(const char* family) {
  static const char* const names[UA2_CNT_N] = {"gemm2", "gemm", "skinny2", "gemv", "rsplit"};
  if (!family) return -1;
  for (int i = 0; i < UA2_CNT_N; ++i)
    if (!strcmp(family, names[i])) return g_ua2_launches[i].load(std::memory_order_relaxed);
  return -1;
}
extern "C" void ua2_debug_refresh_env(void) { g_ua2_env_gen.fetch_add(1, std::memory_order_acq_rel); }

extern "C" size_t ua2_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(ua2_kv_geom);
    case 1: return sizeof(ua2_linear_args);
    case 2: return sizeof(ua2_attn_args);
    case 3: return sizeof(ua2_conv1d_args);
    case 4: return sizeof(ua2_gpt_desc);
    case 5: return sizeof(ua2_stage3_desc);
    case 6: return sizeof(ua2_convtc_args);
    default: return 0;
  }
}

namespace {

// ---- producer half of the scaled-norm hand-over for row-wise kernels (include/ua2hip.h ua2_handover) ----------------
// The sum of squares of a 16-column tile must be added in the tree the linear epilogues use (ua2_linear_common.h
// ssq_tile16: butterfly xor 1, 2, 4, 8 over 16 lanes holding one column each).  A thread holding 8 (4) consecutive columns
// does the first three (two) levels in registers — same operands, same pairing, the first level fused as ssq_tile16 spells it
// — and the rest by exchanging with its neighbour thread(s): the bits are those lane 0 of the 16-lane butterfly ends with.
__device__ __forceinline__ void handover_emit8(const ua2_handover& ho, const float (&v)[8], int m, int c, int C) {
  // first level as ssq_tile16 spells it: the even column's fma(v, v, RN(odd neighbour^2))
  const float p01 = __fmaf_rn(v[0], v[0], __fmul_rn(v[1], v[1])), p23 = __fmaf_rn(v[2], v[2], __fmul_rn(v[3], v[3]));
  const float p45 = __fmaf_rn(v[4], v[4], __fmul_rn(v[5], v[5])), p67 = __fmaf_rn(v[6], v[6], __fmul_rn(v[7], v[7]));
  const float b0 = __fadd_rn(p01, p23), b1 = __fadd_rn(p45, p67);
  float s = __fadd_rn(b0, b1);
  s = __fadd_rn(s, __shfl_xor(s, 1));                       // the other half of the tile lives in the neighbour thread
  if (((c >> 3) & 1) == 0) ho.ssq[(size_t)m * (C >> 4) + (c >> 4)] = s;
  const float4 w0 = *reinterpret_cast<const float4*>(ho.norm_w + c), w1 = *reinterpret_cast<const float4*>(ho.norm_w + c + 4);
  u32x4 pk;
  pk[0] = (unsigned)f2bf(__fmul_rn(v[0], w0.x)) | ((unsigned)f2bf(__fmul_rn(v[1], w0.y)) << 16);
  pk[1] = (unsigned)f2bf(__fmul_rn(v[2], w0.z)) | ((unsigned)f2bf(__fmul_rn(v[3], w0.w)) << 16);
  pk[2] = (unsigned)f2bf(__fmul_rn(v[4], w1.x)) | ((unsigned)f2bf(__fmul_rn(v[5], w1.y)) << 16);
  pk[3] = (unsigned)f2bf(__fmul_rn(v[6], w1.z)) | ((unsigned)f2bf(__fmul_rn(v[7], w1.w)) << 16);
  if (ho.h) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(ho.h) + (size_t)m * ho.ldh + c) = pk;
  if (ho.packed) {   // 8 consecutive k of one chunk = the 16 bytes of lane (k % 32) / 8 * 16 + (m & 15)
    const size_t slot = (((size_t)(m >> 4) * (C >> 5) + (c >> 5)) * 64 + ((c & 31) >> 3) * 16 + (m & 15));
    reinterpret_cast<u32x4*>(ho.packed)[slot] = pk;
  }
}
__device__ __forceinline__ void handover_emit4(const ua2_handover& ho, const float4& v, int m, int c, int C) {
  float s = __fadd_rn(__fmaf_rn(v.x, v.x, __fmul_rn(v.y, v.y)), __fmaf_rn(v.z, v.z, __fmul_rn(v.w, v.w)));
  s = __fadd_rn(s, __shfl_xor(s, 1));                       // columns c ^ 4
  s = __fadd_rn(s, __shfl_xor(s, 2));                       // columns c ^ 8
  if ((c & 15) == 0) ho.ssq[(size_t)m * (C >> 4) + (c >> 4)] = s;
  const float4 w = *reinterpret_cast<const float4*>(ho.norm_w + c);
  uint2 pk;
  pk.x = (unsigned)f2bf(__fmul_rn(v.x, w.x)) | ((unsigned)f2bf(__fmul_rn(v.y, w.y)) << 16);
  pk.y = (unsigned)f2bf(__fmul_rn(v.z, w.z)) | ((unsigned)f2bf(__fmul_rn(v.w, w.w)) << 16);
  if (ho.h) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(ho.h) + (size_t)m * ho.ldh + c) = pk;
  if (ho.packed) {
    const size_t slot = (((size_t)(m >> 4) * (C >> 5) + (c >> 5)) * 64 + ((c & 31) >> 3) * 16 + (m & 15));
    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ho.packed) + slot * 16 + (c & 7) * 2) = pk;
  }
}

// model_new.py:594-600 (_embed_audio_tokens + masked sum over the 8 streams), :604 (wte)
// One workgroup per row; a thread owns 8 consecutive channels (16 B of a bf16 table row, 32 B of an fp32
// one).  The token ids are read once, then all ncb + 1 table rows are requested before the first is summed:
// the gathers are independent HBM misses and must overlap (issued one by one they cost ~1 us each).
template <int DT, int NCB>
__global__ __launch_bounds__(512) void embed_frame_kernel(int C, int ncb, int va, const int32_t* __restrict__ tokens,
                                                          const uint8_t* __restrict__ mask, const void* __restrict__ audio_emb,
                                                          const void* __restrict__ wte, float* __restrict__ audio_sum,
                                                          float* __restrict__ text, const ua2_handover ho) {
  const int m = blockIdx.x;
  const int32_t* tk = tokens + (size_t)m * (ncb + 1);
  const uint8_t* mk = mask + (size_t)m * (ncb + 1);
  int32_t id[NCB + 1];
  bool on[NCB];
#pragma unroll
  for (int i = 0; i <= NCB; ++i) id[i] = tk[i];
#pragma unroll
  for (int i = 0; i < NCB; ++i) on[i] = mk[i] != 0;
  for (int c = threadIdx.x * 8; c < C; c += blockDim.x * 8) {
    float e[NCB + 1][8];
#pragma unroll
    for (int i = 0; i <= NCB; ++i) {
      const void* tab = i < NCB ? audio_emb : wte;
      const size_t row = i < NCB ? (size_t)id[i] + (size_t)i * va : (size_t)id[NCB];
      if constexpr (DT == UA2_BF16) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(tab) + row * C + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          e[i][2 * q] = bf2f((unsigned short)(raw[q] & 0xffffu));
          e[i][2 * q + 1] = bf2f((unsigned short)(raw[q] >> 16));
        }
      } else {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(tab) + row * C + c);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(tab) + row * C + c + 4);
        e[i][0] = a.x; e[i][1] = a.y; e[i][2] = a.z; e[i][3] = a.w;
        e[i][4] = b.x; e[i][5] = b.y; e[i][6] = b.z; e[i][7] = b.w;
      }
    }
    float s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      s[q] = 0.f;
#pragma unroll
      for (int i = 0; i < NCB; ++i) s[q] += on[i] ? e[i][q] : 0.f;   // (audio_embeds * mask).sum(dim=2): i = 0..7 in order
    }
    float* as = audio_sum + (size_t)m * C + c;
    float* tx = text + (size_t)m * C + c;
    *reinterpret_cast<float4*>(as) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(as + 4) = make_float4(s[4], s[5], s[6], s[7]);
    *reinterpret_cast<float4*>(tx) = make_float4(e[NCB][0], e[NCB][1], e[NCB][2], e[NCB][3]);
    *reinterpret_cast<float4*>(tx + 4) = make_float4(e[NCB][4], e[NCB][5], e[NCB][6], e[NCB][7]);
    if constexpr (DT == UA2_BF16) {
      if (ho.norm_w) handover_emit8(ho, s, m, c, C);     // the understanding expert's first layer reads audio_sum (C % 16 == 0: pairs of threads)
    }
  }
}

// generic codebook count (scalar; the model has 8)
template <int DT>
__global__ void embed_frame_any_kernel(int C, int ncb, int va, const int32_t* __restrict__ tokens,
                                       const uint8_t* __restrict__ mask, const void* __restrict__ audio_emb,
                                       const void* __restrict__ wte, float* __restrict__ audio_sum,
                                       float* __restrict__ text) {
  const int m = blockIdx.x;
  const int32_t* tk = tokens + (size_t)m * (ncb + 1);
  const uint8_t* mk = mask + (size_t)m * (ncb + 1);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < ncb; ++i) {
      const float e = load_elem<DT>(audio_emb, ((size_t)tk[i] + (size_t)i * va) * C + c);
      s += mk[i] ? e : 0.f;
    }
    audio_sum[(size_t)m * C + c] = s;
    text[(size_t)m * C + c] = load_elem<DT>(wte, (size_t)tk[ncb] * C + c);
  }
}

// lit_model.py:883-890 (ln_f, :164) + model_new.py:607,610,613 blends
// One workgroup of 256 threads per row, float4 per thread per step, the row kept in registers between the
// statistic and the scaling (C <= 4096; wider rows are re-read).  Fixed summation order: thread-local fma
// chain over its float4s in ascending order, xor-shuffle tree, four wave partials left to right.
__global__ __launch_bounds__(256) void rmsnorm_blend_kernel(int C, const float* __restrict__ x, const float* __restrict__ w, float eps,
                                                            const float* __restrict__ other, const uint8_t* __restrict__ mask, int mask_ld,
                                                            int col_a, int col_b, float* __restrict__ out1, float* __restrict__ out2,
                                                            const ua2_handover ho) {
  constexpr int KEEP = 4;
  __shared__ float part[4];
  const int m = blockIdx.x;
  const float* xr = x + (size_t)m * C;
  const float fa = (col_a >= 0) ? (float)mask[(size_t)m * mask_ld + col_a] : 1.f;
  const float fb = other ? (float)mask[(size_t)m * mask_ld + col_b] : 0.f;
  float4 keep[KEEP];
  float ss = 0.f;
  int it = 0;
  for (int c = threadIdx.x * 4; c < C; c += 1024, ++it) {
    const float4 t = *reinterpret_cast<const float4*>(xr + c);
    if (it < KEEP) keep[it] = t;
    ss = __fmaf_rn(t.x, t.x, ss); ss = __fmaf_rn(t.y, t.y, ss); ss = __fmaf_rn(t.z, t.z, ss); ss = __fmaf_rn(t.w, t.w, ss);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = ((part[0] + part[1]) + part[2]) + part[3];
  const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
  it = 0;
  for (int c = threadIdx.x * 4; c < C; c += 1024, ++it) {
    const float4 t = it < KEEP ? keep[it] : *reinterpret_cast<const float4*>(xr + c);
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    float4 n, o1;
    n.x = __fmul_rn(__fmul_rn(t.x, rstd), wv.x); n.y = __fmul_rn(__fmul_rn(t.y, rstd), wv.y);
    n.z = __fmul_rn(__fmul_rn(t.z, rstd), wv.z); n.w = __fmul_rn(__fmul_rn(t.w, rstd), wv.w);
    o1.x = __fmul_rn(n.x, fa); o1.y = __fmul_rn(n.y, fa); o1.z = __fmul_rn(n.z, fa); o1.w = __fmul_rn(n.w, fa);
    if (other) {
      const float4 ov = *reinterpret_cast<const float4*>(other + (size_t)m * C + c);
      o1.x = __fadd_rn(o1.x, __fmul_rn(ov.x, fb)); o1.y = __fadd_rn(o1.y, __fmul_rn(ov.y, fb));
      o1.z = __fadd_rn(o1.z, __fmul_rn(ov.z, fb)); o1.w = __fadd_rn(o1.w, __fmul_rn(ov.w, fb));
    }
    *reinterpret_cast<float4*>(out1 + (size_t)m * C + c) = o1;
    if (out2) *reinterpret_cast<float4*>(out2 + (size_t)m * C + c) = n;
    if (ho.norm_w) handover_emit4(ho, o1, m, c, C);       // out1 enters the next GPT's first layer (C % 16 == 0: whole groups of 4 threads)
  }
}

// model_new.py:146-187 at topk=1 (masked arg-max; lowest index on ties) + :640,662-663 (_embed_audio)
template <int DT>
__global__ void argmax_embed_kernel(int n_part, const float* __restrict__ pmax, const int32_t* __restrict__ pidx,
                                    int32_t* __restrict__ out_tokens, int out_ld, int out_col,
                                    const void* __restrict__ emb, int emb_off, int C, float* __restrict__ next_h) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int tok_s;
  const int m = blockIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int p = threadIdx.x; p < n_part; p += blockDim.x) {
    const float v = pmax[(size_t)m * n_part + p];
    const int i = pidx[(size_t)m * n_part + p];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    tok_s = bi;
    out_tokens[(size_t)m * out_ld + out_col] = bi;
  }
  __syncthreads();
  if (emb) {
    const size_t row = (size_t)tok_s + (size_t)emb_off;
    for (int c = threadIdx.x; c < C; c += blockDim.x) next_h[(size_t)m * C + c] = load_elem<DT>(emb, row * C + c);
  }
}


// Round 6 — the same arg-max, fused with a gather from the PROJECTED embedding table of the frame executor (ua2_stage3.hip): inside
// the depth decoder's loop (model_new.py:630-641) step i + 1 starts with self.projection(ci_embed), and ci_embed = _embed_audio(i,
// ci_sample) is a row of a fixed table — projection(row), and the scaled-norm hand-over of it (RNE_bf16(y (.) w_norm), per-16-column
// sums of squares), are functions of the sampled id alone.  The executor builds them once per plan WITH THE SAME LAUNCHES the frame
// would run (a row's bits do not depend on the rows beside it: the row-invariance contract), and the frame gathers: y -> next_x [M, Cd];
// hand-over rows -> ho.h (row-major) or ho.packed (fragment order), ho.ssq.  Seven 5.9-us GEMVs per B = 1 frame disappear.
__global__ __launch_bounds__(256) void argmax_gather_kernel(int n_part, const float* __restrict__ pmax, const int32_t* __restrict__ pidx,
                                                            int32_t* __restrict__ out_tokens, int out_ld, int out_col,
                                                            const float* __restrict__ tab_y, const unsigned short* __restrict__ tab_h,
                                                            const float* __restrict__ tab_ssq, long long row_off, int Cd,
                                                            float* __restrict__ next_x, ua2_handover ho, ua2_qkv_gather qg) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int tok_s;
  const int m = blockIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int p = threadIdx.x; p < n_part; p += blockDim.x) {
    const float v = pmax[(size_t)m * n_part + p];
    const int i = pidx[(size_t)m * n_part + p];
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
    tok_s = bi;
    out_tokens[(size_t)m * out_ld + out_col] = bi;
  }
  __syncthreads();
  const size_t row = (size_t)((long long)tok_s + row_off);
  const float4* y4 = reinterpret_cast<const float4*>(tab_y + row * Cd);
  float4* x4 = reinterpret_cast<float4*>(next_x + (size_t)m * Cd);
  for (int c = threadIdx.x; c < Cd / 4; c += blockDim.x) x4[c] = y4[c];
  if (qg.tab_q) {     // layer 0's q | k | v of the next step (position qg.pos): q -> the executor's q buffer, k / v -> this sequence's cache page
    const float4* q4 = reinterpret_cast<const float4*>(qg.tab_q + row * qg.qn);
    float4* qo = reinterpret_cast<float4*>(qg.q_out + (size_t)m * qg.qn);
    for (int c = threadIdx.x; c < qg.qn / 4; c += blockDim.x) qo[c] = q4[c];
    const int hs = qg.kv.head_size, kvw = qg.kv.n_kv * hs;                 // elements of k (v) per row
    const int page = qg.kv.page_table[(size_t)m * qg.kv.max_pages + ua2_page_slot(qg.kv, qg.pos)];
    const int per16 = 16 / qg.esz;                                          // elements per 16-byte piece
    for (int c = threadIdx.x; c < kvw / per16; c += blockDim.x) {
      const int e0 = c * per16, kvh = e0 / hs, dd = e0 - kvh * hs;
      const size_t dst = ((((size_t)page * qg.kv.n_kv + kvh) * UA2_PAGE + (qg.pos % UA2_PAGE)) * hs + dd) * qg.esz;
      const size_t src = (row * kvw + e0) * qg.esz;
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(qg.kv.k_pool) + dst) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(qg.tab_k) + src);
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(qg.kv.v_pool) + dst) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(qg.tab_v) + src);
    }
  }
  if (!ho.ssq) return;
  const int np = Cd >> 4;
  for (int p = threadIdx.x; p < np; p += blockDim.x) ho.ssq[(size_t)m * np + p] = tab_ssq[row * np + p];
  const uint2* h4 = reinterpret_cast<const uint2*>(tab_h + row * Cd);     // four bf16 per piece
  for (int c = threadIdx.x; c < Cd / 4; c += blockDim.x) {
    const uint2 v = h4[c];
    if (ho.h) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(ho.h) + (size_t)m * ho.ldh + 4 * c) = v;
    if (ho.packed) {       // fragment order [M/16][Cd/32][64 lanes][8 bf16]: columns 4c .. 4c + 3 are four consecutive elements of one lane's piece
      const int k = 4 * c, ch = k >> 5, r = k & 31, g = r >> 3, e = r & 7;
      const size_t elem = (((size_t)(m >> 4) * (Cd >> 5) + ch) * 64 + g * 16 + (m & 15)) * 8 + e;
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(ho.packed) + elem) = v;
    }
  }
}

// model_new.py:618-622, 634-637: classifier-free guidance over the (conditional, unconditional) logit rows
//   guided = l1 + (l0 - l1) * scale      (each operation rounded once, as torch evaluates it)
// Both rows are overwritten with `guided`, and the per-16-column arg-max partials of both rows are rebuilt
// from it (same format and tie rule as the UA2_EPI_STORE epilogue), so the sampling tail runs unchanged and
// gives both rows the same token.
__global__ __launch_bounds__(256) void cfg_mix_kernel(float* __restrict__ logits, int ld, int V, float scale,
                                                      const int32_t* __restrict__ forbid, float* __restrict__ pmax,
                                                      int32_t* __restrict__ pidx) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nb = (V + 15) / 16;
  const int pair = blockIdx.y;                             // rows 2 pair, 2 pair + 1
  logits += (size_t)2 * pair * ld;
  pmax += (size_t)2 * pair * nb;
  pidx += (size_t)2 * pair * nb;
  const int fb = forbid ? forbid[2 * pair] : 0;
  float g = -INFINITY;
  if (n < V) {
    const float l0 = logits[n], l1 = logits[ld + n];
    g = __fadd_rn(l1, __fmul_rn(__fsub_rn(l0, l1), scale));
    logits[n] = g;
    logits[ld + n] = g;
  }
  float bv = (n < V && n >= fb) ? g : -INFINITY;
  int bi = n;
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 15) == 0 && n < V) {
    pmax[n / 16] = bv; pidx[n / 16] = bi;
    pmax[nb + n / 16] = bv; pidx[nb + n / 16] = bi;
  }
}

}  // namespace

extern "C" int ua2_embed_frame(int dtype, int32_t M, int32_t C, int32_t n_cb, int32_t va, const int32_t* tokens,
                               const uint8_t* mask, const void* audio_emb, const void* wte, float* audio_sum,
                               float* text, const ua2_handover* hop, void* stream) {
  UA2_CHECK(M > 0 && C > 0 && tokens && mask && audio_emb && wte && audio_sum && text, "ua2_embed_frame: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const bool fast = n_cb == 8 && C % 8 == 0;
  ua2_handover ho{};
  if (hop && hop->norm_w) {
    UA2_CHECK(dtype == UA2_BF16 && fast && C % 32 == 0 && hop->ssq && (hop->h || hop->packed) && (!hop->h || hop->ldh % 8 == 0),
              "ua2_embed_frame: hand-over needs UA2_BF16, n_cb == 8, C %% 32 == 0, ssq and h (ldh %% 8 == 0) or packed");
    ho = *hop;
  }
  const int nthr = std::min(512, std::max(64, (C / 8 + 63) / 64 * 64));
  if (dtype == UA2_BF16 && fast)
    hipLaunchKernelGGL((embed_frame_kernel<UA2_BF16, 8>), dim3(M), dim3(nthr), 0, s, C, n_cb, va, tokens, mask, audio_emb, wte, audio_sum, text, ho);
  else if (dtype == UA2_F32 && fast)
    hipLaunchKernelGGL((embed_frame_kernel<UA2_F32, 8>), dim3(M), dim3(nthr), 0, s, C, n_cb, va, tokens, mask, audio_emb, wte, audio_sum, text, ho);
  else if (dtype == UA2_BF16)
    hipLaunchKernelGGL((embed_frame_any_kernel<UA2_BF16>), dim3(M), dim3(256), 0, s, C, n_cb, va, tokens, mask, audio_emb, wte, audio_sum, text);
  else if (dtype == UA2_F32)
    hipLaunchKernelGGL((embed_frame_any_kernel<UA2_F32>), dim3(M), dim3(256), 0, s, C, n_cb, va, tokens, mask, audio_emb, wte, audio_sum, text);
  else {
    ua2_set_error("ua2_embed_frame: bad dtype %d", dtype);
    return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_rmsnorm_blend(int32_t M, int32_t C, const float* x, const float* w, float eps, const float* other,
                                 const uint8_t* mask, int32_t mask_ld, int32_t col_a, int32_t col_b, float* out1,
                                 float* out2, const ua2_handover* hop, void* stream) {
  UA2_CHECK(M > 0 && C > 0 && C % 4 == 0 && x && w && out1, "ua2_rmsnorm_blend: bad arguments (C must be a multiple of 4)");
  UA2_CHECK((col_a < 0 && !other) || mask, "ua2_rmsnorm_blend: mask needed");
  ua2_handover ho{};
  if (hop && hop->norm_w) {
    UA2_CHECK(C % 32 == 0 && hop->ssq && (hop->h || hop->packed) && (!hop->h || hop->ldh % 8 == 0),
              "ua2_rmsnorm_blend: hand-over needs C %% 32 == 0, ssq and h (ldh %% 8 == 0) or packed");
    ho = *hop;
  }
  hipLaunchKernelGGL(rmsnorm_blend_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, C, x, w, eps, other, mask,
                     mask_ld, col_a, col_b, out1, out2, ho);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_argmax_embed(int dtype, int32_t M, int32_t n_part, const float* part_max, const int32_t* part_idx,
                                int32_t* out_tokens, int32_t out_ld, int32_t out_col, const void* emb,
                                int32_t emb_row_offset, int32_t C, float* next_h, void* stream) {
  UA2_CHECK(M > 0 && n_part > 0 && part_max && part_idx && out_tokens, "ua2_argmax_embed: bad arguments");
  UA2_CHECK(!emb || next_h, "ua2_argmax_embed: next_h is NULL");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == UA2_BF16)
    hipLaunchKernelGGL((argmax_embed_kernel<UA2_BF16>), dim3(M), dim3(256), 0, s, n_part, part_max, part_idx, out_tokens, out_ld, out_col, emb, emb_row_offset, C, next_h);
  else if (dtype == UA2_F32)
    hipLaunchKernelGGL((argmax_embed_kernel<UA2_F32>), dim3(M), dim3(256), 0, s, n_part, part_max, part_idx, out_tokens, out_ld, out_col, emb, emb_row_offset, C, next_h);
  else {
    ua2_set_error("ua2_argmax_embed: bad dtype %d", dtype);
    return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

// rows of a scratch cache (row r = page r, written at position pos[r] by a q|k|v launch) -> compact table rows [n][n_kv * head_size]
__global__ void kv_rows_extract_kernel(const char* __restrict__ k_pool, const char* __restrict__ v_pool, const int32_t* __restrict__ pos, int n,
                                       int n_kv, int hs, int esz, char* __restrict__ out_k, char* __restrict__ out_v) {
  const size_t roww = (size_t)n_kv * hs * esz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * roww; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / roww, b = i - r * roww, kvh = b / ((size_t)hs * esz), w = b - kvh * hs * esz;
    const size_t src = (((r * n_kv + kvh) * UA2_PAGE + (pos[r] % UA2_PAGE)) * hs) * esz + w;
    out_k[i] = k_pool[src];
    out_v[i] = v_pool[src];
  }
}
int ua2_kv_rows_extract(const void* k_pool, const void* v_pool, const int32_t* pos, int n, int n_kv, int hs, int esz, void* out_k, void* out_v, hipStream_t s) {
  hipLaunchKernelGGL(kv_rows_extract_kernel, dim3(std::min(1024, n * 4)), dim3(256), 0, s, (const char*)k_pool, (const char*)v_pool, pos, n, n_kv, hs, esz, (char*)out_k, (char*)out_v);
  UA2_LAUNCH_CHECK();
  return 0;
}

int ua2_argmax_gather(int32_t M, int32_t n_part, const float* part_max, const int32_t* part_idx, int32_t* out_tokens, int32_t out_ld,
                      int32_t out_col, const float* tab_y, const void* tab_h, const float* tab_ssq, int64_t row_off, int32_t Cd, float* next_x,
                      const ua2_handover* ho, const ua2_qkv_gather* qg, hipStream_t s) {
  UA2_CHECK(M > 0 && n_part > 0 && part_max && part_idx && out_tokens && tab_y && next_x && Cd % 32 == 0, "ua2_argmax_gather: bad arguments");
  ua2_qkv_gather q{};
  if (qg) {
    UA2_CHECK(qg->tab_q && qg->tab_k && qg->tab_v && qg->q_out && qg->qn % 4 == 0 && (qg->esz == 2 || qg->esz == 4) && qg->kv.k_pool && qg->kv.v_pool &&
                  qg->kv.page_table && qg->kv.head_size % (16 / qg->esz) == 0 && qg->kv.ring_pages == 0,
              "ua2_argmax_gather: q | k | v table / destination missing");
    q = *qg;
  }
  ua2_handover h{};
  if (ho) {
    UA2_CHECK(tab_h && tab_ssq && ho->ssq && (ho->h || ho->packed) && (!ho->h || ho->ldh % 4 == 0), "ua2_argmax_gather: hand-over tables / outputs missing");
    h = *ho;
  }
  hipLaunchKernelGGL(argmax_gather_kernel, dim3(M), dim3(256), 0, s, n_part, part_max, part_idx, out_tokens, out_ld, out_col, tab_y,
                     reinterpret_cast<const unsigned short*>(tab_h), tab_ssq, (long long)row_off, Cd, next_x, h, q);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_cfg_mix(float* logits, int32_t ld, int32_t V, float scale, const int32_t* forbid, float* part_max,
                           int32_t* part_idx, int32_t pairs, void* stream) {
  UA2_CHECK(logits && part_max && part_idx && V > 0 && ld >= V && pairs > 0 && pairs < 65536, "ua2_cfg_mix: bad arguments");
  hipLaunchKernelGGL(cfg_mix_kernel, dim3((V + 255) / 256, pairs), dim3(256), 0, (hipStream_t)stream, logits, ld, V, scale, forbid,
                     part_max, part_idx);
  UA2_LAUNCH_CHECK();
  return 0;
}
