// Short-context attention of the local (depth) decoder: at codebook step i a row attends to the i + 1 <= 8
// positions its own frame has written so far (model_new.py:629-641 through lit_model.py:468-481, 529-531).
// Internal, shared by the stand-alone kernel (ua2_attn.hip: ua2_attn_local) and by the decode kernel's
// UA2_PRO_LOCAL_ATTN prologue (ua2_gemv.hip), which must agree bit for bit: a single row runs fused into the
// O-projection, several rows run the stand-alone kernel, and a row's ids may not depend on which.
//
// One wave computes 128 / HS heads at a time; a lane owns two consecutive dims of one head.  All K and V
// rows of the context are requested up front (<= 16 small loads, one L2 round trip), scores are reduced with
// DPP inside the HS/2 lanes of a head, softmax and the weighted sum of V run in position order with explicit
// single-rounding operations (no contraction), so the result is a pure function of (q row, cache contents).
#pragma once
#include "ua2_common.h"

constexpr int kLocalCtx = 8;   // positions a row may attend to (audio_num_codebooks of the model)

template <int CTRL>
__device__ __forceinline__ float ua2_dpp_add(float v) {
  return __fadd_rn(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)));
}
// all-reduce (sum) over aligned groups of LPR lanes, LPR in {16, 32, 64}
template <int LPR>
__device__ __forceinline__ float ua2_group_sum(float v) {
  v = ua2_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = ua2_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = ua2_dpp_add<0x141>(v);   // row_half_mirror
  v = ua2_dpp_add<0x140>(v);   // row_mirror
  if constexpr (LPR >= 32) v = __fadd_rn(v, __shfl_xor(v, 16));
  if constexpr (LPR >= 64) v = __fadd_rn(v, __shfl_xor(v, 32));
  return v;
}

template <int DT, int HS>
struct LocalAttn {
  static constexpr int LPH = HS / 2;        // lanes per head
  static constexpr int HPW = 64 / LPH;      // heads per wave pass
  static constexpr int BYTES = Elem<DT>::BYTES;
  float2 q;
  float2 k[kLocalCtx], v[kLocalCtx];

  // issue every load of head `h` (this lane's dims d, d + 1); `page` = the sequence's first cache page
  __device__ __forceinline__ void issue(const ua2_kv_geom& kv, const float* __restrict__ q_row, int page, int h, int d) {
    const int G = kv.n_head / kv.n_kv;
    q = *reinterpret_cast<const float2*>(q_row + (size_t)h * HS + d);
    const size_t base = (((size_t)page * kv.n_kv + h / G) * UA2_PAGE) * HS + d;
#pragma unroll
    for (int j = 0; j < kLocalCtx; ++j) {
      if constexpr (DT == UA2_BF16) {
        const unsigned kr = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(kv.k_pool) + base + (size_t)j * HS);
        const unsigned vr = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(kv.v_pool) + base + (size_t)j * HS);
        k[j] = make_float2(__uint_as_float(kr << 16), __uint_as_float(kr & 0xffff0000u));
        v[j] = make_float2(__uint_as_float(vr << 16), __uint_as_float(vr & 0xffff0000u));
      } else {
        k[j] = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(kv.k_pool) + base + (size_t)j * HS);
        v[j] = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(kv.v_pool) + base + (size_t)j * HS);
      }
    }
  }
  // softmax(q k^T / sqrt(HS)) v over positions 0..pos; returns this lane's two output dims
  __device__ __forceinline__ float2 finish(int pos) const {
    const float scale = 1.0f / sqrtf((float)HS);
    const float qx = __fmul_rn(q.x, scale), qy = __fmul_rn(q.y, scale);
    float s[kLocalCtx];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < kLocalCtx; ++j) {
      const float d = ua2_group_sum<LPH>(__fmaf_rn(qx, k[j].x, __fmul_rn(qy, k[j].y)));
      s[j] = (j <= pos) ? d : -INFINITY;
      m = fmaxf(m, s[j]);
    }
    float l = 0.f, ox = 0.f, oy = 0.f;
#pragma unroll
    for (int j = 0; j < kLocalCtx; ++j) {
      const float p = (j <= pos) ? __builtin_amdgcn_exp2f(__fmul_rn(__fsub_rn(s[j], m), 1.44269504088896340736f)) : 0.f;
      l = __fadd_rn(l, p);
      ox = __fmaf_rn(p, v[j].x, ox);
      oy = __fmaf_rn(p, v[j].y, oy);
    }
    const float inv = 1.0f / l;
    return make_float2(__fmul_rn(ox, inv), __fmul_rn(oy, inv));
  }
};
