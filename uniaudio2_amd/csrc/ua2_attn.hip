// Paged GQA attention for single-position query rows (decode, and prefill row by row).
//
// Replaces lit_model.py:468-481 (slice the cache to input_pos_maxp1, repeat_interleave K/V to
// n_head copies) + :529-531 (masked F.scaled_dot_product_attention).  The causal mask of the
// reference ("key position <= query position", build_mask_cache :863-866 gathered by
// input_pos :137) is applied as a length: row r attends to positions 0..row_pos[r].
//
// MI355X design: one workgroup per (row, kv-head) walks positions 0..row_pos once; the K and V rows
// are read with 16-byte lane loads and shared by the q_per_kv query heads of the group (no repeat_interleave
// copies); the summation order depends only on the position, never on the batch.
#include <algorithm>

#include "ua2_common.h"
#include "ua2_attn_local.h"

namespace {

constexpr int kMaxG = 4;  // query heads per kv head (Llama-3.2-3B: 3, local decoder: 4)

// ---- single-pass variant: one workgroup per (row, kv-head) --------------------------------------
// After the first profiles (profiles/r1_a, r1_b): at B = 1 attention is pure latency, so
//   * the row's positions 0..pos are split evenly over the 8 waves (not by page), each wave walks
//     its range 4*UNR rows at a time with all K and V loads of a step issued together;
//   * the 16-lane dot-product reductions use DPP row operations (no LDS permutes);
//   * exp is v_exp_f32 (exp2 of a pre-scaled argument);
//   * online softmax per wave, the <= 8 wave states merge through LDS in wave order, so the
//     summation order depends only on the position (batch invariant, deterministic).
// Output is the normalised attention row in fp32 (input of the O-projection).
constexpr int kFusedWaves = 8;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// all-reduce (sum) over aligned groups of LPR lanes, LPR in {4, 8, 16, 32}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v = dpp_add<0xB1>(v);                          // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                          // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v = dpp_add<0x141>(v);  // row_half_mirror
  if constexpr (LPR >= 16) v = dpp_add<0x140>(v); // row_mirror
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
  return v;
}
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

template <int DT, int HS>
__global__ __launch_bounds__(kFusedWaves * 64) void attn_fused_kernel(const ua2_attn_args a) {
  constexpr int EPL = Elem<DT>::EPL, BYTES = Elem<DT>::BYTES;
  constexpr int LPR = HS / EPL, RPW = 64 / LPR;   // lanes per cache row, row groups per wave
  constexpr int UNR = (DT == UA2_BF16) ? 4 : 2;   // wave instructions per step (K and V each)
  constexpr int NS = kFusedWaves * RPW;           // independent online-softmax states per workgroup
  // Every 16-/8-lane row group keeps its own (m, l, o) state: no cross-group traffic inside the
  // loop (the first version spent ~6 us in serialized ds_bpermute chains).  States merge once,
  // through LDS, in (wave, group) order.
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int r = blockIdx.x, kvh = blockIdx.y;
  const int G = a.kv.n_head / a.kv.n_kv;
  float* st_m = sm;                    // [NS][G]
  float* st_l = st_m + NS * kMaxG;     // [NS][G]
  float* st_o = st_l + NS * kMaxG;     // [NS][G][HS]

  const int pos = a.row_pos[r];
  const int seq = a.row_seq ? a.row_seq[r] : r;   // NULL: row r is sequence r (decode batches)
  const int n = pos + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPR, rin = lane / LPR;
  const float scale = 1.0f / sqrtf((float)HS);
  const int32_t* ptab = a.kv.page_table + (size_t)seq * a.kv.max_pages;
  // contiguous range of this wave, a multiple of RPW rows
  const int lo = (a.window > 0) ? max(0, n - a.window) : 0;   // Moshi `context`: delta < context (transformer.py:405-406)
  const int chunk = ((n - lo + kFusedWaves * RPW - 1) / (kFusedWaves * RPW)) * RPW;
  const int j0 = lo + wave * chunk, j1 = min(n, j0 + chunk);

  float q[kMaxG][EPL];
#pragma unroll
  for (int h = 0; h < kMaxG; ++h) {
    const float* qp = a.q + ((size_t)r * a.kv.n_head + (size_t)kvh * G + (h < G ? h : 0)) * HS + sub * EPL;
    const float sc = (h < G) ? scale : 0.f;
#pragma unroll
    for (int e4 = 0; e4 < EPL / 4; ++e4) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 4 * e4);
      q[h][4 * e4 + 0] = t.x * sc; q[h][4 * e4 + 1] = t.y * sc; q[h][4 * e4 + 2] = t.z * sc; q[h][4 * e4 + 3] = t.w * sc;
    }
  }
  float m_run[kMaxG], l_run[kMaxG], o_run[kMaxG][EPL];
#pragma unroll
  for (int h = 0; h < kMaxG; ++h) {
    m_run[h] = -INFINITY;
    l_run[h] = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o_run[h][e] = 0.f;
  }

  // K/V loads of step t+1 are issued before the arithmetic of step t (one step of prefetch)
  auto issue = [&](int jb, u32x4 (&kr)[UNR], u32x4 (&vr)[UNR]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = jb + u * RPW + rin;
      const int jc = (j < j1) ? j : j0;           // clamp: unconditional loads, masked in the math
      const size_t off = ((((size_t)ptab[jc / UA2_PAGE] * a.kv.n_kv + kvh) * UA2_PAGE + (jc % UA2_PAGE)) * HS +
                          (size_t)sub * EPL) * BYTES;
      kr[u] = *reinterpret_cast<const u32x4*>((const char*)a.kv.k_pool + off);
      vr[u] = *reinterpret_cast<const u32x4*>((const char*)a.kv.v_pool + off);
    }
  };
  u32x4 kraw[UNR], vraw[UNR], knext[UNR], vnext[UNR];
  if (j0 < j1) issue(j0, kraw, vraw);
  for (int jb = j0; jb < j1; jb += UNR * RPW) {
    const bool more = jb + UNR * RPW < j1;
    if (more) issue(jb + UNR * RPW, knext, vnext);
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) ok[u] = (jb + u * RPW + rin) < j1;
    float s[UNR][kMaxG];
    float gmax[kMaxG];
#pragma unroll
    for (int h = 0; h < kMaxG; ++h) gmax[h] = -INFINITY;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float kf[EPL];
      if constexpr (DT == UA2_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kf[2 * e] = __uint_as_float(kraw[u][e] << 16);
          kf[2 * e + 1] = __uint_as_float(kraw[u][e] & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) kf[e] = __uint_as_float(kraw[u][e]);
      }
#pragma unroll
      for (int h = 0; h < kMaxG; ++h) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) d += q[h][e] * kf[e];
        d = group_sum<LPR>(d);
        s[u][h] = ok[u] ? d : -INFINITY;
        gmax[h] = fmaxf(gmax[h], s[u][h]);
      }
    }
    // online softmax of THIS row group (its rows jb + u*RPW + rin, u < UNR); u = 0 may be masked
    // for trailing groups, so guard the all-masked case
#pragma unroll
    for (int h = 0; h < kMaxG; ++h) {
      const float m_new = fmaxf(m_run[h], gmax[h]);
      const float resc = (m_new == -INFINITY) ? 1.f : fast_exp(m_run[h] - m_new);
      m_run[h] = m_new;
      l_run[h] *= resc;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o_run[h][e] *= resc;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float vf[EPL];
      if constexpr (DT == UA2_BF16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vf[2 * e] = __uint_as_float(vraw[u][e] << 16);
          vf[2 * e + 1] = __uint_as_float(vraw[u][e] & 0xffff0000u);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) vf[e] = __uint_as_float(vraw[u][e]);
      }
#pragma unroll
      for (int h = 0; h < kMaxG; ++h) {
        const float p = ok[u] ? fast_exp(s[u][h] - m_run[h]) : 0.f;
        l_run[h] += p;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o_run[h][e] += p * vf[e];
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) { kraw[u] = knext[u]; vraw[u] = vnext[u]; }
    }
  }
  // publish the state of this row group
  const int sidx = wave * RPW + rin;
#pragma unroll
  for (int h = 0; h < kMaxG; ++h) {
    if (h < G) {
      if (sub == 0) { st_m[sidx * kMaxG + h] = m_run[h]; st_l[sidx * kMaxG + h] = l_run[h]; }
      float* o = st_o + ((size_t)sidx * G + h) * HS + sub * EPL;
#pragma unroll
      for (int e4 = 0; e4 < EPL / 4; ++e4)
        *reinterpret_cast<float4*>(o + 4 * e4) =
            make_float4(o_run[h][4 * e4], o_run[h][4 * e4 + 1], o_run[h][4 * e4 + 2], o_run[h][4 * e4 + 3]);
    }
  }
  __syncthreads();
  // merge, phase 1: wave h turns the NS (m, l) pairs of head h into normalised weights
  //   wgt[w] = exp(m_w - M) / sum_w exp(m_w - M) l_w        (0 for groups that saw no row)
  float* wgt = st_l;   // overwrite l in place
  if (wave < G) {
    const int h = wave;
    constexpr int SPL = (NS + 63) / 64;   // states per lane
    float mw[SPL], lw[SPL];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int w = lane + 64 * k;
      mw[k] = (w < NS) ? st_m[w * kMaxG + h] : -INFINITY;
      lw[k] = (w < NS) ? st_l[w * kMaxG + h] : 0.f;
      mx = fmaxf(mx, mw[k]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float f[SPL], den = 0.f;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      f[k] = (mw[k] == -INFINITY) ? 0.f : fast_exp(mw[k] - mx);
      den += f[k] * lw[k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) den += __shfl_xor(den, o);
    const float inv = 1.0f / den;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int w = lane + 64 * k;
      if (w < NS) wgt[w * kMaxG + h] = f[k] * inv;
    }
  }
  __syncthreads();
  // phase 2: y[h][d] = sum_w wgt[w][h] * o[w][h][d], fixed (wave, group) order
  for (int idx = tid; idx < G * HS; idx += kFusedWaves * 64) {
    const int h = idx / HS, d = idx - h * HS;
    float acc = 0.f;
#pragma unroll 8
    for (int w = 0; w < NS; ++w) acc += wgt[w * kMaxG + h] * st_o[((size_t)w * G + h) * HS + d];
    if (a.y) a.y[((size_t)r * a.kv.n_head + (size_t)kvh * G + h) * HS + d] = acc;
    if (a.y_packed) store_packed_operand<DT>(a.y_packed, r, (kvh * G + h) * HS + d, a.kv.n_head * HS / Elem<DT>::KC, acc);
  }
}

template <int DT, int HS>
void launch_fused_hs(const ua2_attn_args& a, hipStream_t s) {
  constexpr int EPL = Elem<DT>::EPL, RPW = 64 / (HS / EPL), NS = kFusedWaves * RPW;
  const int G = a.kv.n_head / a.kv.n_kv;
  const size_t smem = (size_t)(2 * NS * kMaxG + (size_t)NS * G * HS) * sizeof(float);
  constexpr auto kern = attn_fused_kernel<DT, HS>;
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, dim3(a.R, a.kv.n_kv), dim3(kFusedWaves * 64), smem, s, a);
}

template <int DT>
int launch_fused(const ua2_attn_args& a, hipStream_t s) {
  switch (a.kv.head_size) {
    case 32: launch_fused_hs<DT, 32>(a, s); break;
    case 64: launch_fused_hs<DT, 64>(a, s); break;
    case 128: launch_fused_hs<DT, 128>(a, s); break;
    default:
      ua2_set_error("ua2_attn: head_size %d not supported (32, 64, 128)", a.kv.head_size);
      return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

// ---- short-context (local decoder) form: one workgroup per row, 128 / HS heads per wave pass ----
template <int DT, int HS>
__global__ __launch_bounds__(1024) void attn_local_kernel(const ua2_attn_args a) {
  using LA = LocalAttn<DT, HS>;
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int pos = a.row_pos[r];
  const int page = a.kv.page_table[(size_t)(a.row_seq ? a.row_seq[r] : r) * a.kv.max_pages];
  const float* q_row = a.q + (size_t)r * a.kv.n_head * HS;
  for (int h0 = wave * LA::HPW; h0 < a.kv.n_head; h0 += nw * LA::HPW) {
    const int h = h0 + lane / LA::LPH, d = (lane % LA::LPH) * 2;
    LA la;
    la.issue(a.kv, q_row, page, h, d);
    const float2 o = la.finish(pos);
    if (a.y) *reinterpret_cast<float2*>(a.y + (size_t)r * a.kv.n_head * HS + (size_t)h * HS + d) = o;
    if (a.y_packed) {
      store_packed_operand<DT>(a.y_packed, r, h * HS + d, a.kv.n_head * HS / Elem<DT>::KC, o.x);
      store_packed_operand<DT>(a.y_packed, r, h * HS + d + 1, a.kv.n_head * HS / Elem<DT>::KC, o.y);
    }
  }
}

template <int DT>
int launch_local(const ua2_attn_args& a, hipStream_t s) {
  const int hpw = 128 / a.kv.head_size;
  const int waves = std::min(16, std::max(1, a.kv.n_head / hpw));
  switch (a.kv.head_size) {
    case 32: hipLaunchKernelGGL((attn_local_kernel<DT, 32>), dim3(a.R), dim3(waves * 64), 0, s, a); break;
    case 64: hipLaunchKernelGGL((attn_local_kernel<DT, 64>), dim3(a.R), dim3(waves * 64), 0, s, a); break;
    case 128: hipLaunchKernelGGL((attn_local_kernel<DT, 128>), dim3(a.R), dim3(waves * 64), 0, s, a); break;
    default:
      ua2_set_error("ua2_attn_local: head_size %d not supported (32, 64, 128)", a.kv.head_size);
      return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int ua2_attn_launch(const ua2_attn_args& a, hipStream_t s) {
  UA2_CHECK(a.R > 0, "ua2_attn: R=%d", a.R);
  UA2_CHECK(a.q && a.row_pos && (a.y || a.y_packed) && a.kv.k_pool && a.kv.v_pool && a.kv.page_table,
            "ua2_attn: NULL pointer argument");
  UA2_CHECK(a.kv.n_kv > 0 && a.kv.n_head % a.kv.n_kv == 0 && a.kv.n_head / a.kv.n_kv <= kMaxG,
            "ua2_attn: n_head=%d n_kv=%d not supported (group size <= %d)", a.kv.n_head, a.kv.n_kv, kMaxG);
  UA2_CHECK(!a.y_packed || (a.kv.n_head * a.kv.head_size) % (a.dtype == UA2_BF16 ? 32 : 16) == 0, "ua2_attn: y_packed needs n_head*head_size %% chunk == 0");
  if (a.dtype == UA2_BF16) return launch_fused<UA2_BF16>(a, s);
  if (a.dtype == UA2_F32) return launch_fused<UA2_F32>(a, s);
  ua2_set_error("ua2_attn: bad dtype %d", a.dtype);
  return -1;
}

int ua2_attn_local_launch(const ua2_attn_args& a, hipStream_t s) {
  UA2_CHECK(a.R > 0 && a.q && a.row_pos && (a.y || a.y_packed) && a.kv.k_pool && a.kv.v_pool && a.kv.page_table, "ua2_attn_local: bad arguments");
  UA2_CHECK(a.kv.n_kv > 0 && a.kv.n_head % a.kv.n_kv == 0 && a.kv.n_head % (128 / std::max(a.kv.head_size, 1)) == 0,
            "ua2_attn_local: n_head=%d n_kv=%d head_size=%d not supported", a.kv.n_head, a.kv.n_kv, a.kv.head_size);
  if (a.dtype == UA2_BF16) return launch_local<UA2_BF16>(a, s);
  if (a.dtype == UA2_F32) return launch_local<UA2_F32>(a, s);
  ua2_set_error("ua2_attn_local: bad dtype %d", a.dtype);
  return -1;
}

extern "C" int ua2_attn_local(const ua2_attn_args* a, void* stream) {
  UA2_CHECK(a != nullptr, "ua2_attn_local: NULL args");
  return ua2_attn_local_launch(*a, (hipStream_t)stream);
}

extern "C" int ua2_attn(const ua2_attn_args* a, void* stream) {
  UA2_CHECK(a != nullptr, "ua2_attn: NULL args");
  return ua2_attn_launch(*a, (hipStream_t)stream);
}
