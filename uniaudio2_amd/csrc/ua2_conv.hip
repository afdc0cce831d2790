// 1-D convolution / transposed convolution of the codec's waveform auto-encoders.  Three kernels behind ua2_conv1d:
//   conv1d_kernel       precision 0: exact fp32 on the f32 matrix pipe (encode side, parity reference)        — first below
//   conv1d_x3_kernel    precision 1: bf16 x 3 split operands on the bf16 matrix pipe, plain phase-by-phase form
//   conv1d_x3p_kernel   precision 1: the same arithmetic (bit-identical), software-pipelined — the default of the decode side
//
// Replaces (§2.3 K14-K16, K19, K21):
//   ReasoningCodec_film/models/scalar24k.py  Conv1d :36-74 (causal = left zero-pad d(k-1), else
//     symmetric), ConvTranspose1d :76-112 (causal: k = 2s, trim the last s), PReLU / tanh /
//     round(9x)/9 / repeat-upsample epilogues (:120,133,136-138,148-149,206,243,266,278,289,385),
//     AudioDiffusion1D.py:188,244-251 strided k = s down-samplers;
//   MimiCodec/model/modules/conv.py StreamingConv1d :232-254 (left pad k_eff - s, extra right pad),
//     StreamingConvTranspose1d :306-329 (trim k - s), seanet.py:92-94 ELU -> conv resblocks.
//
// One kernel does all of them as an implicit GEMM on the f32-input matrix pipe
// (v_mfma_f32_16x16x4_f32: exact fp32 multiply-add, 1/16 of the bf16 rate but 2.4x a VALU conv and
// bit-comparable with an fma chain).  With the codec's channel counts (32..1024) the arithmetic
// intensity is ~100 flop/byte, so the bound is the f32 MFMA pipe, not HBM; activations make exactly
// one trip through HBM per layer because bias, activation, residual add, input pre-activation,
// repeat-upsampling and the transposed conv's phase interleave are fused into the load / store.
//
//   out[n][t] = act( bias[n] + sum_{ci,j} W[n][ci*K + j] * pre(x[ci][(t*stride + j*dil - pad_left) / in_repeat]) ) (+ residual)
//   transposed conv with stride P: P phase-convolutions, packed rows n = phase*Cout + co, written to
//   y[co][t*P + phase - out_trim_left]  (the taps of a phase are W[ci][co][phase + m*P], m descending).
//
// Tiling: workgroup = 4 waves = 64 output rows x 16*NTT time steps (NTT = 4, 2 or 1: the launcher shrinks the time
// tile until the grid has >= 2 workgroups per CU — the 512-channel layers at 12.5 / 50 Hz have few time steps and ran
// at 15 % of the f32 MFMA peak on 192 workgroups); wave w owns rows 16w..16w+15 and NTT 16-wide time tiles.  Input channels are consumed 16 at a time: their window
// (zero-padded, pre-activated) is staged once in LDS and feeds K chunks of 16 reduction indices; the
// weight fragments come pre-tiled (ua2_pack_linear fp32 layout over [rows][Cin_pad*K]) as one 16-byte
// load per lane per chunk, reused by the 16 MFMAs of the four time tiles.
#include "ua2_common.h"
#include <string.h>

namespace {

constexpr int kBR = 64;       // output rows per workgroup
constexpr int kCIG = 16;      // input channels per staging group
constexpr int kMaxK = 32;

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  switch (act) {
    case UA2_ACT_PRELU: return v >= 0.f ? v : __fmul_rn(alpha, v);   // single roundings: hipcc would otherwise contract a*v + c differently per kernel
    case UA2_ACT_ELU: return v > 0.f ? v : expm1f(v);               // nn.ELU(alpha=1)
    case UA2_ACT_TANH: return tanhf(v);
    case UA2_ACT_ROUND9: return rintf(9.f * v) / 9.f;               // torch.round(9*x)/9, scalar24k.py:289
    default: return v;
  }
}

template <int NTT>
__global__ __launch_bounds__(256) void conv1d_kernel(const ua2_conv1d_args a) {
  constexpr int kBT = 16 * NTT;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.K, s = a.stride, d = a.dilation;
  const int W = (kBT - 1) * s + (K - 1) * d + 1;       // staged window per input channel
  const int Wp = W + 1;                                // +1: de-phase the rows across LDS banks
  float* xs = sm;                                      // [kCIG][Wp]
  int* koff = reinterpret_cast<int*>(xs + kCIG * Wp);  // [kCIG*K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 15, g = lane >> 4;
  const int t0 = blockIdx.x * kBT;
  const int r0 = blockIdx.y * kBR + wave * 16;
  const int b = blockIdx.z;
  const int rows = a.Cout * a.out_phases;
  const int cin_pad = (a.Cin + kCIG - 1) / kCIG * kCIG;
  const int nchunks = cin_pad * K / 16;                // packed chunks per row tile
  const int tin_eff = a.Tin * a.in_repeat;

  for (int kk = tid; kk < kCIG * K; kk += 256) koff[kk] = (kk / K) * Wp + (kk % K) * d;

  f32x4 acc[NTT];
#pragma unroll
  for (int nt = 0; nt < NTT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_active = r0 < rows;
  const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + (size_t)(r0 / 16) * nchunks * 64 + lane;
  const float pre_alpha = (a.pre_act == UA2_ACT_PRELU && a.pre_alpha) ? a.pre_alpha[0] : 0.f;
  const int in_start = t0 * s - a.pad_left;

  for (int cg = 0; cg < cin_pad / kCIG; ++cg) {
    __syncthreads();
    for (int idx = tid; idx < kCIG * W; idx += 256) {
      const int cl = idx / W, wi = idx - cl * W;
      const int ci = cg * kCIG + cl, ti = in_start + wi;
      float v = 0.f;
      if (ci < a.Cin && ti >= 0 && ti < tin_eff) {
        v = a.x[((size_t)b * a.Cin + ci) * a.Tin + ti / a.in_repeat];
        v = apply_act(v, a.pre_act, pre_alpha);
      }
      xs[cl * Wp + wi] = v;
    }
    __syncthreads();
    if (wave_active) {
      for (int c = 0; c < K; ++c) {
        const f32x4 wa = __builtin_bit_cast(f32x4, wp[(size_t)(cg * K + c) * 64]);
        const int4 ko = *reinterpret_cast<const int4*>(&koff[c * 16 + g * 4]);
        const int kov[4] = {ko.x, ko.y, ko.z, ko.w};
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const int tb = (nt * 16 + tl) * s;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], xs[kov[e] + tb], acc[nt], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_active) return;
  // epilogue: D[row = (lane>>4)*4 + r][col = lane & 15]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = r0 + g * 4 + r;
    if (n >= rows) continue;
    const int phase = n / a.Cout, co = n - phase * a.Cout;
    const float bias = a.bias ? a.bias[co] : 0.f;
    const float alpha = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? co : 0] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      const int t = t0 + nt * 16 + tl;
      const int to = t * a.out_phases + phase - a.out_trim_left;
      if (to < 0 || to >= a.Tout) continue;
      float v = __fadd_rn(acc[nt][r], bias);
      v = apply_act(v, a.post_act, alpha);
      const size_t o = ((size_t)b * a.Cout + co) * a.Tout + to;
      if (a.residual) v += a.residual[o];
      a.y[o] = v;
    }
  }
}

// ---- bf16 x 3 form ("precision = 1") --------------------------------------------------------------------
// The exact-fp32 kernel above is bound by the f32 matrix pipe (1/16 of the bf16 rate), 4x slower than the bytes it
// moves.  Here every fp32 operand is split into two bf16 halves (hi = RNE(x), lo = RNE(x - hi): 16 significant
// bits together) and the product is taken as  Wh Xh + Wh Xl + Wl Xh  on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation: 3 MFMAs at 16x the rate = 5.3x the f32 pipe; the dropped Wl Xl term and the split's own rounding are
// ~2^-16 relative per product (measured end to end against the goldens: DESIGN.md §5).  The exact kernel stays the
// reference for that measurement and serves the encode side (its latents feed integer decisions).
//
// Layout.  K index of the implicit GEMM = (channel group of 32, tap j, channel within the group): one MFMA K-chunk
// is ONE tap over 32 input channels.  The input window of a channel group is staged in LDS time-major:
// plane[w][32 channels] of bf16 (hi plane, lo plane; 80-byte rows: 64 B of channels + 16 B pad so that neither the
// transposing writes nor the 16-byte reads collide on banks), so the B operand of a (tap, time tile) is one
// ds_read_b128 per lane per plane: lane (g = lane >> 4, t = lane & 15) reads channels 8g..8g+7 at window position
// (t0 + t) * stride + j * dilation.  No per-element address tables, no integer division in the staging loop.
// Weights come pre-split and pre-tiled ([rows/16][chunks][64 lanes][16 B], hi and lo buffers) straight from L2, one
// chunk ahead of the MFMAs.  A wave owns 16 output rows x NTT time tiles; the four waves of a workgroup cover
// `rt` row tiles x (4 / rt) time sub-blocks, so layers with 32 or 16 output rows still use every wave.
constexpr int kCG3 = 32;       // channels per staging group
constexpr int kRowB = 80;      // LDS bytes per window position per plane

template <int NTT, int RPW>
__global__ __launch_bounds__(256, 4) void conv1d_x3_kernel(const ua2_conv1d_args a, const int rt) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const int K = a.K, s = a.stride, d = a.dilation;
  const int tsub = 4 / rt;
  constexpr int kBT = 16 * NTT;                        // time steps per wave
  const int wgt = kBT * tsub;                          // time steps per workgroup
  const int W = (wgt - 1) * s + (K - 1) * d + 1;       // staged window positions
  char* xh = smc;
  char* xl = smc + (size_t)W * kRowB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 15, g = lane >> 4;
  const int wr = wave % rt, wt = wave / rt;
  const int t0 = blockIdx.x * wgt;
  const int tw0 = wt * kBT;
  const int r0 = (blockIdx.y * rt + wr) * (16 * RPW);  // RPW row tiles per wave: every x fragment read from LDS feeds RPW x 3 MFMAs
  const int b = blockIdx.z;
  const int rows = a.Cout * a.out_phases;
  const int ngroups = (a.Cin + kCG3 - 1) / kCG3;
  const int nchunks = ngroups * K;
  const int tin_eff = a.Tin * a.in_repeat;
  const bool wave_active = r0 < rows;
  const int ntile_rows = (rows + 15) / 16;
  const u32x4* wph[RPW];
  const u32x4* wpl[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int tile = min(r0 / 16 + q, ntile_rows - 1);  // a wave's second tile may lie past the last row tile: clamped load, masked store
    wph[q] = reinterpret_cast<const u32x4*>(a.w) + (size_t)tile * nchunks * 64 + lane;
    wpl[q] = reinterpret_cast<const u32x4*>(a.w_lo) + (size_t)tile * nchunks * 64 + lane;
  }
  const float pre_alpha = (a.pre_act == UA2_ACT_PRELU && a.pre_alpha) ? a.pre_alpha[0] : 0.f;
  const int in_start = t0 * s - a.pad_left;

  f32x4 acc[RPW][NTT];
#pragma unroll
  for (int q = 0; q < RPW; ++q)
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging walk of this thread: (channel pair p, window position wi), advanced by 256 elements per step without a division
  const int p_step = 256 / W, w_step = 256 - p_step * W;
  const int p_first = tid / W, w_first = tid - p_first * W;
  u32x4 wh[RPW], wl[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) { wh[q] = u32x4{0u, 0u, 0u, 0u}; wl[q] = wh[q]; }
  if (wave_active) {
#pragma unroll
    for (int q = 0; q < RPW; ++q) { wh[q] = wph[q][0]; wl[q] = wpl[q][0]; }
  }

  // Residual values of this wave's output tile, requested up front: read in the epilogue one by one (each load in front of a
  // store the compiler must assume it aliases) they cost a memory round trip apiece — 32 of them, ~30 us of a 40 us launch.
  float resv[RPW][4][NTT];
#pragma unroll
  for (int q = 0; q < RPW; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        resv[q][r][nt] = 0.f;
        if (a.residual) {
          const int n = r0 + q * 16 + g * 4 + r;
          const int phase = n / a.Cout, co = n - phase * a.Cout;
          const int to = (t0 + tw0 + nt * 16 + tl) * a.out_phases + phase - a.out_trim_left;
          if (n < rows && to >= 0 && to < a.Tout) resv[q][r][nt] = a.residual[((size_t)b * a.Cout + co) * a.Tout + to];
        }
      }

  for (int cg = 0; cg < ngroups; ++cg) {
    __syncthreads();
    // All global loads of a batch go out before the first conversion: issued one by one behind their LDS writes they
    // cost a full memory round trip each (~20 per workgroup: the first version of this kernel was bound by exactly that).
    constexpr int SB = 8;
    for (int p = p_first, wi = w_first; p < kCG3 / 2; ) {
      float v0[SB], v1[SB];
      int pp[SB], ww[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        pp[u] = p; ww[u] = wi;
        v0[u] = 0.f; v1[u] = 0.f;
        const int ci = cg * kCG3 + 2 * p, ti = in_start + wi;
        if (p < kCG3 / 2 && ti >= 0 && ti < tin_eff) {
          const int tsrc = (a.in_repeat == 1) ? ti : (a.in_repeat == 2 ? (ti >> 1) : ti / a.in_repeat);
          const size_t off = ((size_t)b * a.Cin + ci) * a.Tin + tsrc;
          if (ci < a.Cin) v0[u] = a.x[off];
          if (ci + 1 < a.Cin) v1[u] = a.x[off + a.Tin];
        }
        p += p_step; wi += w_step;
        if (wi >= W) { wi -= W; ++p; }
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        if (pp[u] < kCG3 / 2) {
          const float x0 = apply_act(v0[u], a.pre_act, pre_alpha), x1 = apply_act(v1[u], a.pre_act, pre_alpha);   // act(0) = 0 for every pre-activation
          const unsigned h0 = f2bf(x0), h1 = f2bf(x1);
          const unsigned l0 = f2bf(__fsub_rn(x0, bf2f((unsigned short)h0))), l1 = f2bf(__fsub_rn(x1, bf2f((unsigned short)h1)));
          *reinterpret_cast<unsigned*>(xh + (size_t)ww[u] * kRowB + pp[u] * 4) = h0 | (h1 << 16);
          *reinterpret_cast<unsigned*>(xl + (size_t)ww[u] * kRowB + pp[u] * 4) = l0 | (l1 << 16);
        }
      }
    }
    __syncthreads();
    if (wave_active) {
      for (int j = 0; j < K; ++j) {
        const int chunk = cg * K + j;
        bf16x8 ah[RPW], al[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) { ah[q] = __builtin_bit_cast(bf16x8, wh[q]); al[q] = __builtin_bit_cast(bf16x8, wl[q]); }
        if (chunk + 1 < nchunks) {                                    // one chunk ahead
#pragma unroll
          for (int q = 0; q < RPW; ++q) { wh[q] = wph[q][(size_t)(chunk + 1) * 64]; wl[q] = wpl[q][(size_t)(chunk + 1) * 64]; }
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const size_t o = (size_t)((tw0 + nt * 16 + tl) * s + j * d) * kRowB + g * 16;
          const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xh + o));
          const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xl + o));
#pragma unroll
          for (int q = 0; q < RPW; ++q) {
            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[q], bh, acc[q][nt], 0, 0, 0);   // small terms first
            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bl, acc[q][nt], 0, 0, 0);
            acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bh, acc[q][nt], 0, 0, 0);
          }
        }
      }
    }
  }
  if (a.w2) {
    // ---- fused residual unit (scalar24k.py:143-151): y = x + act2(W2 act1(conv1(x) + b1) + b2), W2 a 1 x 1 conv ----
    // The workgroup holds ALL C output channels of its time tile (launcher: gridDim.y == 1), so the 1 x 1 conv's reduction
    // over channels closes inside it: the intermediate h never goes to HBM (un-fused: written, re-read, plus a second
    // launch — 5 passes over a C x T tensor per unit instead of 2, on layers that are HBM-bound).
    const int C = a.Cout, ng2 = C / kCG3;
    char* hbase = smc + 2 * (size_t)W * kRowB;                          // [ng2][2 planes][wgt][kRowB]
    const size_t hplane = (size_t)wgt * kRowB;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int nb = r0 + q * 16 + g * 4;                               // this lane's 4 consecutive channels nb .. nb+3
      const int grp = nb / kCG3, pc = nb % kCG3;
      float bs[4], al1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bs[r] = a.bias ? a.bias[nb + r] : 0.f;
        al1[r] = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? nb + r : 0] : 0.f;
      }
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const int tw = tw0 + nt * 16 + tl;
        unsigned hh[4], hl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = apply_act(__fadd_rn(acc[q][nt][r], bs[r]), a.post_act, al1[r]);
          hh[r] = f2bf(hv);
          hl[r] = f2bf(__fsub_rn(hv, bf2f((unsigned short)hh[r])));
        }
        char* dst = hbase + (size_t)grp * 2 * hplane + (size_t)tw * kRowB + pc * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
        *reinterpret_cast<uint2*>(dst + hplane) = make_uint2(hl[0] | (hl[1] << 16), hl[2] | (hl[3] << 16));
      }
    }
    __syncthreads();
    f32x4 acc2[RPW][NTT];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) acc2[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cg = 0; cg < ng2; ++cg) {
      bf16x8 ah[RPW], al[RPW];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const size_t wo = ((size_t)(r0 / 16 + q) * ng2 + cg) * 64 + lane;
        ah[q] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(a.w2)[wo]);
        al[q] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(a.w2_lo)[wo]);
      }
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const size_t o = (size_t)cg * 2 * hplane + (size_t)(tw0 + nt * 16 + tl) * kRowB + g * 16;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(hbase + o));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(hbase + o + hplane));
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          acc2[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[q], bh, acc2[q][nt], 0, 0, 0);
          acc2[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bl, acc2[q][nt], 0, 0, 0);
          acc2[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bh, acc2[q][nt], 0, 0, 0);
        }
      }
    }
    const float alpha2 = a.alpha2 ? a.alpha2[0] : 0.f;
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = r0 + q * 16 + g * 4 + r;
        const float b2 = a.bias2 ? a.bias2[n] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const int t = t0 + tw0 + nt * 16 + tl;
          if (t >= a.Tout) continue;
          const size_t o = ((size_t)b * C + n) * a.Tout + t;
          float v = __fadd_rn(acc2[q][nt][r], b2);
          v = v >= 0.f ? v : __fmul_rn(alpha2, v);
          a.y[o] = __fadd_rn(v, resv[q][r][nt]);
        }
      }
    return;
  }
  if (!wave_active) return;
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = r0 + q * 16 + g * 4 + r;
      if (n >= rows) continue;
      const int phase = n / a.Cout, co = n - phase * a.Cout;
      const float bias = a.bias ? a.bias[co] : 0.f;
      const float alpha = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? co : 0] : 0.f;
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const int t = t0 + tw0 + nt * 16 + tl;
        const int to = t * a.out_phases + phase - a.out_trim_left;
        if (to < 0 || to >= a.Tout) continue;
        float v = __fadd_rn(acc[q][nt][r], bias);
        v = apply_act(v, a.post_act, alpha);
        const size_t o = ((size_t)b * a.Cout + co) * a.Tout + to;
        a.y[o] = __fadd_rn(v, resv[q][r][nt]);
      }
    }
  }
}

// ---- bf16 x 3, software-pipelined ---------------------------------------------------------------------------
// conv1d_x3_kernel above runs its phases back to back inside a workgroup — request a channel group's window, wait a
// memory round trip, convert, barrier, MFMAs with the weights one chunk (~100 cycles of MFMA) ahead of an ~600-cycle
// L2 hit, epilogue — and leans on 3 co-resident workgroups to fill the gaps.  Measured per phase (UA2_CONV_DBG
// experiments, profiles/r2_notes.md): the phases add up almost linearly; the 512-channel layers at T = 1500 (16
// channel groups x a round trip each) ran at 1/10 of either roofline.  Here a workgroup walks a sequence of UNITS
// (time tile, run of `gpu` channel groups) and every global load is consumed one whole unit after it was issued:
//   * the x window of unit u+1 is requested right after the barrier that opens unit u, rides in registers (<= 16
//     (channel pair, position) elements per thread) through u's MFMAs, and is converted + written to the OTHER LDS image;
//   * all weight chunks of a unit (<= 8 of 16 B per lane per operand half) live in registers; chunk c of unit u+1 is
//     requested into the registers of chunk c the moment u's MFMAs have read them.  vmcnt retires in order, so the
//     x request goes out BEFORE those refills: the wait in front of the conversion then covers only the x loads, and
//     each refill is first waited for one unit later;
//   * a workgroup takes `tpw` consecutive time tiles, so the layers with one or two channel groups (C = 32, 64: all of
//     the 120 / 240 kHz-rate work) pipeline across tiles.
// One barrier per unit (two more inside a fused residual-unit epilogue).  Arithmetic, operand split and summation
// order are those of conv1d_x3_kernel: the two kernels are bit-identical (tests/test_gpu_conv.py).
constexpr int kKC = 8;         // most weight chunks (tap x channel group) a unit holds in registers

// NTT 16-step time tiles per wave (one 16-row tile per wave), CPU weight chunks per unit = GPU channel groups x K taps.
//
// Staging map.  A unit's window is GPU x 16 channel pairs x W positions.  Wave w owns the 4 * GPU pairs
// 4*GPU*w .. 4*GPU*(w+1)-1 and walks the window 64 positions at a time: element k of a thread is (position group k / PPW,
// pair k % PPW) — both compile-time — so the channel row is wave-uniform (scalar base address, scalar validity), a wave-load
// is 256 contiguous bytes, and the per-lane index work (clamp, repeat-upsampling division, padding mask, LDS row) is done
// once per position group, not per element.  (The first pipelined version walked (pair, position) with per-element index
// arithmetic: ~1400 VALU instructions per unit against 84 MFMAs — VALU-bound at 6 us per unit.)
//
// The unit loop is straight-line on purpose — the request for the unit after the last one is clamped onto the last, its
// conversion lands in the idle LDS image — so that the s_waitcnt the compiler places in front of the conversion counts
// exactly the weight refills issued behind the x loads, and nothing pending crosses the loop's back edge except those.
template <int NTT, int CPU, int GPU, int NPG, bool UPT1, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv1d_x3p_kernel(const ua2_conv1d_args a, const int rt, const int tpw, const int ntiles,
                                                                               const int nrb, const int ngx) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  constexpr int K = CPU / GPU;
  constexpr int PPW = 16 * GPU / NW;                    // channel pairs per wave (NW = 4 waves, or 8 for the fused 128-channel unit)
  constexpr int SB = PPW * NPG;                         // NPG position groups of 64 staged per unit (launcher: W <= 64 * NPG)
  const int s = a.stride, d = a.dilation;
  const int tsub = NW / rt;
  constexpr int kBT = 16 * NTT;
  const int wgt = kBT * tsub;
  const int W = (wgt - 1) * s + (K - 1) * d + 1;
  // LDS image rows are padded to whole position groups: every lane of a staged group writes a row (rows >= W are never
  // read) and a group past the window (NPG is the next instantiated count) goes to a dump area, so neither request nor
  // conversion carries a predicate.  (With a lane predicate the compiler wraps conversion + s_waitcnt in an `execz` skip;
  // on that path the loads stay formally pending and the next write to their registers — at the loop top — becomes an
  // s_waitcnt vmcnt(0) that also drains the tile's stores and the weight refills.  With a uniform `skip this group`
  // branch around the loads, the phi copies of the loaded registers wait for each load right behind its issue.)
  const int npg = (W + 63) >> 6;
  const unsigned planeB = (unsigned)(npg * 64) * kRowB, groupB = 2 * planeB, bufB = (unsigned)GPU * groupB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, g = lane >> 4;
  const int wr = wave % rt, wt = wave / rt;
  const int tw0 = wt * kBT;
  // Workgroup id -> (row block, tile run, batch), row block fastest: ids are dealt round-robin to the 8 XCDs, so an XCD's
  // L2 only ever sees the weight slices of the row blocks congruent to it.  With the row block in blockIdx.y every XCD
  // streamed ALL row blocks' weights through its 4 MB L2: the 512-channel layers (7.3 MB of split weights) fetched 86 MB
  // per launch from the fabric; 31 MB with this mapping (PMC FETCH_SIZE, profiles/r2_pmc_codec.txt).  No change in time:
  // those layers are bound by per-unit latency (16 small units per tile), not by that traffic.
  const int wg_rb = blockIdx.x % nrb, wg_rest = blockIdx.x / nrb;
  const int wg_tx = wg_rest % ngx, b = wg_rest / ngx;
  const int r0 = (wg_rb * rt + wr) * 16;
  const int rows = a.Cout * a.out_phases;
  const int ngroups = (a.Cin + kCG3 - 1) / kCG3;
  const int nchunks = ngroups * K;
  const int upt = ngroups / GPU;                        // units per tile (launcher: GPU divides the channel-group count)
  const int tin_eff = a.Tin * a.in_repeat;
  const int ntile_rows = (rows + 15) / 16;
  const int tile_first = wg_tx * tpw;
  if (tile_first >= ntiles) return;
  const int wtile = min(r0 / 16, ntile_rows - 1);       // a wave past the last row tile computes on a clamped tile and stores nothing
  const u32x4* wph = reinterpret_cast<const u32x4*>(a.w) + (size_t)wtile * nchunks * 64 + lane;
  const u32x4* wpl = reinterpret_cast<const u32x4*>(a.w_lo) + (size_t)wtile * nchunks * 64 + lane;
  const float pre_neg = (a.pre_act == UA2_ACT_PRELU && a.pre_alpha) ? a.pre_alpha[0] : 1.f;   // launcher: pre_act is PReLU or none (slope 1)
  const float* xbat = a.x + (size_t)b * a.Cin * a.Tin;
  const unsigned hbytes = a.w2 ? (unsigned)(a.Cout / kCG3) * 2 * wgt * kRowB : 0u;   // fused unit: the h image sits behind the two x images
  const unsigned rep_magic = a.in_repeat > 1 ? (unsigned)(0x100000000ull / (unsigned)a.in_repeat + 1) : 0u;   // ti / in_repeat = umulhi(ti, magic)

  u32x4 wregh[CPU], wregl[CPU];
#pragma unroll
  for (int c = 0; c < CPU; ++c) { wregh[c] = wph[(size_t)c * 64]; wregl[c] = wpl[(size_t)c * 64]; }

  float v0[SB], v1[SB];
  auto request = [&](int tile, int ug) {                // clamped (always valid) addresses; padding is applied at conversion
    const int in_start = tile * wgt * s - a.pad_left;
    const int c0 = ug * GPU * kCG3 + 2 * PPW * wave;    // this wave's first channel
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) {
      const int ti = min(max(in_start + lane + 64 * pg, 0), tin_eff - 1);
      const int tsrc = rep_magic ? (int)__umulhi((unsigned)ti, rep_magic) : ti;
#pragma unroll
      for (int pi = 0; pi < PPW; ++pi) {
        const float* row0 = xbat + (size_t)min(c0 + 2 * pi, a.Cin - 1) * a.Tin;
        const float* row1 = xbat + (size_t)min(c0 + 2 * pi + 1, a.Cin - 1) * a.Tin;
        v0[pg * PPW + pi] = row0[tsrc];
        v1[pg * PPW + pi] = row1[tsrc];
      }
    }
  };
  auto commit = [&](char* buf, int tile, int ug) {      // zero padding, pre-activation, hi/lo split, LDS image
    const int in_start = tile * wgt * s - a.pad_left;
    const int c0 = ug * GPU * kCG3 + 2 * PPW * wave;
    char* wbase = buf + (unsigned)((PPW * wave) >> 4) * groupB + ((PPW * wave) & 15) * 4;   // this wave's group image, first pair column
    char* dump = smc + 2 * bufB + hbytes + ((PPW * wave) & 15) * 4;                            // [2 planes][64 rows] past the images
#pragma unroll
    for (int pg = 0; pg < NPG; ++pg) {
      const int wi = lane + 64 * pg, ti = in_start + wi;
      const bool pos_ok = ti >= 0 && ti < tin_eff;
      const bool live = pg < npg;                       // uniform
      char* dst = live ? wbase + (unsigned)wi * kRowB : dump + (unsigned)lane * kRowB;
      const unsigned lo_off = live ? planeB : 64u * kRowB;
#pragma unroll
      for (int pi = 0; pi < PPW; ++pi) {
        float x0 = (pos_ok && c0 + 2 * pi < a.Cin) ? v0[pg * PPW + pi] : 0.f;
        float x1 = (pos_ok && c0 + 2 * pi + 1 < a.Cin) ? v1[pg * PPW + pi] : 0.f;
        x0 = x0 >= 0.f ? x0 : __fmul_rn(pre_neg, x0);
        x1 = x1 >= 0.f ? x1 : __fmul_rn(pre_neg, x1);
        unsigned hi, lo;
        split_pair(x0, x1, hi, lo);
        *reinterpret_cast<unsigned*>(dst + pi * 4) = hi;
        *reinterpret_cast<unsigned*>(dst + pi * 4 + lo_off) = lo;
      }
    }
  };

  f32x4 acc[NTT];
  unsigned bofs[NTT];                                   // this lane's B-fragment offset per time tile, tap 0 of group 0
#pragma unroll
  for (int nt = 0; nt < NTT; ++nt) {
    acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bofs[nt] = (unsigned)((tw0 + nt * 16 + tl) * s) * kRowB + g * 16;
  }
  const bool has_res = a.residual != nullptr;
  const float* resp = has_res ? a.residual : a.x;       // branch-free residual: without one, a valid address and a zero select
  // per-row epilogue constants, loaded once: left inside the tile epilogue the compiler schedules these loads into the
  // MFMA phase and the (conditional) epilogue leaves them pending over the loop's back edge — a vmcnt(0) at the loop top
  unsigned yo[4];                                       // launcher: B * Cout * Tout < 2^31
  int ph[4];
  float bias[4], alpha[4], bias2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = min(r0 + g * 4 + r, rows - 1);
    const int phase = n / a.Cout, co = n - phase * a.Cout;
    ph[r] = phase - a.out_trim_left;
    yo[r] = (unsigned)((b * a.Cout + co) * a.Tout);
    bias[r] = a.bias ? a.bias[co] : 0.f;
    alpha[r] = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? co : 0] : 1.f;   // launcher: PReLU or none (slope 1)
    bias2[r] = (a.w2 && a.bias2) ? a.bias2[co] : 0.f;
  }
  const float alpha2 = (a.w2 && a.alpha2) ? a.alpha2[0] : 0.f;
  const bool rows_ok = r0 + 16 <= rows;
  // fused unit: the 1 x 1 conv's weights (C <= 64 with 4 waves, 128 with 8: at most kNG2 chunks per row tile), loaded once — requested inside the
  // epilogue they were waited for on the spot, an L2 round trip per tile
  constexpr int kNG2 = NW == 8 ? 4 : 2;
  u32x4 w2h[kNG2], w2l[kNG2];
#pragma unroll
  for (int cg = 0; cg < kNG2; ++cg) {
    w2h[cg] = u32x4{0u, 0u, 0u, 0u};
    w2l[cg] = w2h[cg];
    if (a.w2 && cg < a.Cout / kCG3) {
      const size_t wo = ((size_t)(r0 / 16) * (a.Cout / kCG3) + cg) * 64 + lane;
      w2h[cg] = reinterpret_cast<const u32x4*>(a.w2)[wo];
      w2l[cg] = reinterpret_cast<const u32x4*>(a.w2_lo)[wo];
    }
  }

  // Loop shape.  vmcnt retires in order, so what a wait costs is decided by what was issued BEFORE the thing waited for:
  //   tile:  residual request (oldest of the tile)
  //     unit:  barrier | x request of the next unit | MFMAs, each chunk refilled in place behind its last use |
  //            conversion of the next unit (waits for its x only: the refills behind it stay in flight)
  //   tile epilogue (stores; a fused residual unit adds LDS image + second GEMM)
  // With one channel-group run per tile (UPT1: every layer at the 120 / 240 kHz rates) the whole body is straight-line
  // and every count the compiler derives is exact.  Loads under a branch the compiler cannot see through — a conditional
  // epilogue inside the unit loop, a predicated conversion, a skipped position group — each ended in an s_waitcnt vmcnt(0)
  // somewhere in the loop during development, draining the prefetch this kernel exists for (profiles/r2_notes.md).
  float resv[4][NTT];
  auto request_residual = [&](int tile) {               // clamped addresses; without a residual a valid address and a zero select
    const int t0 = tile * wgt + tw0 + tl;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const int to = min(max((t0 + nt * 16) * a.out_phases + ph[r], 0), a.Tout - 1);
        resv[r][nt] = resp[has_res ? yo[r] + to : 0];
      }
  };
  int u = 0;                                            // units done: parity = LDS image
  auto unit = [&](int tile_n, int ug_n) {
    ua2_lds_barrier();
    const char* xb = smc + (unsigned)(u & 1) * bufB;
    request(tile_n, ug_n);
    const size_t chunk_n = (size_t)ug_n * CPU;
#pragma unroll
    for (int c = 0; c < CPU; ++c) {
      const unsigned cofs = (unsigned)(c / K) * groupB + (unsigned)((c % K) * d) * kRowB;   // group image, tap
      bf16x8 bh[NTT], bl[NTT];
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const char* src = xb + cofs + bofs[nt];
        bh[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src));
        bl[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + planeB));
      }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, wregh[c]), al = __builtin_bit_cast(bf16x8, wregl[c]);
      // per accumulator the order is al*bh, ah*bl, ah*bh (small terms first, as in conv1d_x3_kernel); across accumulators
      // the MFMAs interleave so that back-to-back issues are independent
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[nt], 0, 0, 0);
      wregh[c] = wph[(chunk_n + c) * 64];               // refill in place: first needed one unit from now
      wregl[c] = wpl[(chunk_n + c) * 64];
    }
    commit(smc + (unsigned)((u + 1) & 1) * bufB, tile_n, ug_n);
    ++u;
  };
  auto epilogue = [&](int tile) {                       // D[row = g*4 + r][col = tl]
    const int t0 = tile * wgt + tw0 + tl;
    const bool interior = rows_ok && a.out_phases == 1 && a.out_trim_left == 0 && tile * wgt + wgt <= a.Tout;   // no store needs a mask
    f32x4 res[NTT];
    if (a.w2) {   // fused residual unit (scalar24k.py:143-151), see conv1d_x3_kernel: h -> LDS image -> 1 x 1 conv -> PReLU -> + x
      const int ng2 = a.Cout / kCG3;
      char* hbase = smc + 2 * bufB;                                       // [ng2][2 planes][wgt][kRowB]
      const unsigned hplane = (unsigned)wgt * kRowB;
      const int nb = r0 + g * 4;
      char* hdst = hbase + (unsigned)(nb / kCG3) * 2 * hplane + (unsigned)(tw0 + tl) * kRowB + (nb % kCG3) * 2;
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = __fadd_rn(acc[nt][r], bias[r]);
          hv[r] = t >= 0.f ? t : __fmul_rn(alpha[r], t);
        }
        unsigned h01, l01, h23, l23;
        split_pair(hv[0], hv[1], h01, l01);
        split_pair(hv[2], hv[3], h23, l23);
        *reinterpret_cast<uint2*>(hdst + nt * 16 * kRowB) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(hdst + nt * 16 * kRowB + hplane) = make_uint2(l01, l23);
      }
      ua2_lds_barrier();
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) res[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cg = 0; cg < kNG2; ++cg) {
        if (cg >= ng2) break;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, w2h[cg]), al = __builtin_bit_cast(bf16x8, w2l[cg]);
        bf16x8 bh[NTT], bl[NTT];
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const char* src = hbase + (unsigned)cg * 2 * hplane + (unsigned)(tw0 + nt * 16 + tl) * kRowB + g * 16;
          bh[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src));
          bl[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(src + hplane));
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) res[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], res[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) res[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], res[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) res[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], res[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __fadd_rn(res[nt][r], bias2[r]);
          res[nt][r] = v >= 0.f ? v : __fmul_rn(alpha2, v);
        }
    } else {
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __fadd_rn(acc[nt][r], bias[r]);
          res[nt][r] = v >= 0.f ? v : __fmul_rn(alpha[r], v);
        }
    }
    if (interior) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) a.y[yo[r] + t0 + nt * 16] = __fadd_rn(res[nt][r], has_res ? resv[r][nt] : 0.f);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const int to = (t0 + nt * 16) * a.out_phases + ph[r];
          if (r0 + g * 4 + r < rows && to >= 0 && to < a.Tout) a.y[yo[r] + to] = __fadd_rn(res[nt][r], has_res ? resv[r][nt] : 0.f);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  const int my_tiles = min(tpw, ntiles - tile_first);
  request(tile_first, 0);
  commit(smc, tile_first, 0);
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile = tile_first + ti;
    const int tile_nx = ti + 1 < my_tiles ? tile + 1 : tile;   // past the last tile: the last unit again, into the idle image
    request_residual(tile);
    if constexpr (UPT1) {
      unit(tile_nx, 0);
    } else {
      for (int ug = 0; ug + 1 < upt; ++ug) unit(tile, ug + 1);
      unit(tile_nx, ti + 1 < my_tiles ? 0 : upt - 1);
    }
    epilogue(tile);
  }
}

template <int NTT, int CPU, int GPU, int NPG, int NW = 4>
void launch_x3p(const ua2_conv1d_args& a, dim3 grid, size_t smem, int rt, int tpw, int ntiles, bool upt1, hipStream_t s) {
  constexpr auto k1 = conv1d_x3p_kernel<NTT, CPU, GPU, NPG, true, NW>;
  constexpr auto kn = conv1d_x3p_kernel<NTT, CPU, GPU, NPG, false, NW>;
  if (upt1) {
    ua2_allow_big_lds<k1>();
    hipLaunchKernelGGL(k1, dim3(grid.x * grid.y * grid.z), dim3(64 * NW), smem, s, a, rt, tpw, ntiles, (int)grid.y, (int)grid.x);
  } else {
    ua2_allow_big_lds<kn>();
    hipLaunchKernelGGL(kn, dim3(grid.x * grid.y * grid.z), dim3(64 * NW), smem, s, a, rt, tpw, ntiles, (int)grid.y, (int)grid.x);
  }
}

template <int NTT, int RPW>
void launch_x3(const ua2_conv1d_args& a, dim3 grid, size_t smem, int rt, hipStream_t s) {
  constexpr auto kern = conv1d_x3_kernel<NTT, RPW>;
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a, rt);
}

__global__ void avgpool1d_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int Tin, int Tout, int k) {
  const int64_t total = rows * Tout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / Tout;
    const int t = (int)(idx - r * Tout);
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += x[r * Tin + (int64_t)t * k + j];
    y[idx] = s / (float)k;   // torch.nn.AvgPool1d(kernel_size=k), scalar24k.py:118
  }
}

// Depthwise (groups == channels) conv / transposed conv of the Mimi resamplers (resample.py:13-119 with
// channel_wise=True: ConvTrUpsample1d in MimiCodec.py:68; k = 2*stride taps per channel).  One output per
// thread, taps in ascending order with fma: a few bytes per flop, HBM-bound and tiny (512 channels x T).
__global__ void dwconv1d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ y, int64_t total, int C, int Tin, int Tout, int K, int stride, int dilation,
                                int pad_left, int transposed) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx % Tout);
    const int64_t bc = idx / Tout;
    const int c = (int)(bc % C);
    const float* xr = x + bc * Tin;
    const float* wr = w + (size_t)c * K;
    float acc = bias ? bias[c] : 0.f;
    if (!transposed) {
      for (int j = 0; j < K; ++j) {
        const int ti = t * stride + j * dilation - pad_left;
        if (ti >= 0 && ti < Tin) acc = __fmaf_rn(wr[j], xr[ti], acc);
      }
    } else {  // y_full[u] = sum_{ti*stride + j == u} w[j] x[ti];  pad_left = samples trimmed on the left
      const int u = t + pad_left;
      for (int j = 0; j < K; ++j) {
        const int r = u - j;
        if (r >= 0 && r % stride == 0 && r / stride < Tin) acc = __fmaf_rn(wr[j], xr[r / stride], acc);
      }
    }
    y[idx] = acc;
  }
}

}  // namespace

extern "C" int ua2_dwconv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t C, int32_t Tin,
                            int32_t Tout, int32_t K, int32_t stride, int32_t dilation, int32_t pad_left, int32_t transposed,
                            void* stream) {
  UA2_CHECK(x && w && y && B > 0 && C > 0 && Tin > 0 && Tout > 0 && K >= 1 && stride >= 1 && dilation >= 1 && pad_left >= 0,
            "ua2_dwconv1d: bad arguments");
  UA2_CHECK(!transposed || dilation == 1, "ua2_dwconv1d: transposed form has no dilation");
  const int64_t total = (int64_t)B * C * Tout;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(dwconv1d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, total, C, Tin, Tout, K,
                     stride, dilation, pad_left, transposed);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_conv1d(const ua2_conv1d_args* a, void* stream) {
  UA2_CHECK(a && a->x && a->w && a->y, "ua2_conv1d: NULL argument");
  UA2_CHECK(a->B > 0 && a->Cin > 0 && a->Cout > 0 && a->Tin > 0 && a->Tout > 0, "ua2_conv1d: empty problem");
  UA2_CHECK(a->K >= 1 && a->K <= kMaxK && a->stride >= 1 && a->dilation >= 1 && a->in_repeat >= 1 && a->out_phases >= 1,
            "ua2_conv1d: bad geometry K=%d stride=%d dil=%d", a->K, a->stride, a->dilation);
  UA2_CHECK(a->out_phases == 1 || (a->stride == 1 && a->dilation == 1), "ua2_conv1d: phase mode needs stride=dilation=1");
  UA2_CHECK(a->post_act != UA2_ACT_PRELU || a->post_alpha, "ua2_conv1d: PReLU needs post_alpha");
  const int tq = a->out_phases == 1 ? a->Tout : ua2_ceil_div(a->Tout + a->out_trim_left, a->out_phases);
  if (a->precision == 1) {
    UA2_CHECK(a->w_lo != nullptr, "ua2_conv1d: precision 1 (bf16 x 3) needs w_lo (ua2 host helper pack_conv_weight_x3)");
    const int rows = a->Cout * a->out_phases;
    const int rpw = rows > 16 ? 2 : 1;                                   // row tiles per wave
    const int wave_rows = 16 * rpw;
    const int rt = rows > 2 * wave_rows ? 4 : (rows > wave_rows ? 2 : 1);  // wave row-groups per workgroup; the other waves split time
    const int row_blocks3 = ua2_ceil_div(rows, wave_rows * rt);
    if (a->w2) {   // fused residual unit: one workgroup must hold every output channel of its time tile
      UA2_CHECK(a->w2_lo && a->residual && a->out_phases == 1 && a->stride == 1 && a->in_repeat == 1 && a->Cin == a->Cout &&
                    (a->Cout == 32 || a->Cout == 64 || a->Cout == 128) && row_blocks3 == 1 && a->Tin == a->Tout,
                "ua2_conv1d: the fused residual unit needs Cin == Cout in {32, 64, 128}, stride 1, Tin == Tout, w2_lo and residual");
    }
    hipStream_t st = (hipStream_t)stream;
    // ---- software-pipelined kernel whenever a unit's weights and window fit its register budget ----
    {
      const int nw = (a->w2 && rows == 128) ? 8 : 4;                     // the fused 128-channel unit: 8 waves hold its 8 row tiles
      int p_rt = rows > 64 && nw == 8 ? 8 : (rows > 32 ? 4 : (rows > 16 ? 2 : 1)), p_ntt = 0, p_tpw = 0, p_gpu = 0;
      if (const char* e = getenv("UA2_CONV_PIPE")) {                     // experiment hook: "ntt,rt,tpw,gpu" (0 = automatic) or "off"
        int v[4] = {0, 0, 0, 0};
        if (!strcmp(e, "off")) p_ntt = -1;
        else if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) {
          p_ntt = v[0]; p_tpw = v[2]; p_gpu = v[3];
          if (!a->w2 && v[1]) p_rt = v[1];
        }
      }
      const int p_rb = ua2_ceil_div(rows, 16 * p_rt);
      const int ngroups = ua2_ceil_div(a->Cin, kCG3);
      // instantiated (K, gpu) pairs and position-group counts; a window needing fewer groups than the next instantiated
      // count sends the excess to a dump area (10 KB)
      auto npg_inst = [&](int gp, int need) {
        if (a->K == 7) return need <= 3 ? need : 0;
        if (gp == 1) return need <= 2 ? 2 : (need <= 4 ? 4 : 0);
        if (gp == 2) return need <= 2 ? need : 0;
        return need == 1 ? 1 : 0;
      };
      auto lds_of = [&](int n, int gp) {
        const int wg = 16 * n * (nw / p_rt);
        const int64_t Wn = (int64_t)(wg - 1) * a->stride + (a->K - 1) * a->dilation + 1;
        const int need = (int)((Wn + 63) / 64), inst = npg_inst(gp, need);
        return 4 * gp * (int64_t)need * 64 * kRowB + (a->w2 ? (int64_t)(a->Cout / 32) * 2 * wg * kRowB : 0) + (inst > need ? 2 * 64 * kRowB : 0);
      };
      auto fits = [&](int n, int gp) {
        const bool inst = (a->K == 7 && gp == 1) || ((a->K == 1 || a->K == 2) && (gp == 1 || gp == 2 || gp == 4));
        if (!inst || ngroups % gp) return false;
        const int wg = 16 * n * (nw / p_rt);
        const int64_t Wn = (int64_t)(wg - 1) * a->stride + (a->K - 1) * a->dilation + 1;
        return npg_inst(gp, (int)((Wn + 63) / 64)) > 0 && lds_of(n, gp) <= (nw == 8 ? 150 : 80) * 1024;   // two workgroups per CU (one of 8 waves)
      };
      const bool slope_acts = (a->pre_act == UA2_ACT_NONE || a->pre_act == UA2_ACT_PRELU) && (a->post_act == UA2_ACT_NONE || a->post_act == UA2_ACT_PRELU);
      const bool small_index = (int64_t)a->B * a->Cin * a->Tin < (1ll << 31) && (int64_t)a->B * a->Cout * a->Tout < (1ll << 31) &&
                               (int64_t)a->Tin * a->in_repeat * a->in_repeat < (1ll << 32);
      int ntt = 4;
      if (p_ntt > 0) ntt = p_ntt;
      else {
        while (ntt > 1 && !fits(ntt, 1)) ntt >>= 1;
        while (ntt > 1 && (int64_t)ua2_ceil_div(tq, 16 * ntt * (nw / p_rt)) * p_rb * a->B < 256) ntt >>= 1;
      }
      const int64_t W0 = (int64_t)(16 * ntt * (nw / p_rt) - 1) * a->stride + (a->K - 1) * a->dilation + 1;
      const bool nw_ok = nw == 4 || (ntt == 4 && a->K == 7 && (W0 + 63) / 64 == 2);    // the one 8-wave instantiation
      if (p_ntt >= 0 && slope_acts && small_index && nw_ok && fits(ntt, 1) && (!a->w2 || p_rb == 1)) {
        int gpu = 1;
        for (int gp = 2; gp <= ngroups && gp <= 4; ++gp)
          if (fits(ntt, gp)) gpu = gp;
        if (p_gpu > 0 && fits(ntt, p_gpu)) gpu = p_gpu;
        const int wgt = 16 * ntt * (nw / p_rt);
        const int ntiles = ua2_ceil_div(tq, wgt);
        const int64_t total = (int64_t)ntiles * p_rb * a->B;
        // consecutive tiles per workgroup: one resident round (2 workgroups per CU) where the layer is long enough — a
        // workgroup's first window is the only one nothing hides (measured on the 120 / 240 kHz layers: 43 -> 35, 33 -> 24.5 us)
        const int slots = nw == 8 ? 256 : 512;
        int tpw = (int)std::min<int64_t>(8, std::max<int64_t>(1, (total + slots - 1) / slots));
        if (p_tpw > 0) tpw = p_tpw;
        const int64_t Wn = (int64_t)(wgt - 1) * a->stride + (a->K - 1) * a->dilation + 1;
        const size_t smem = (size_t)lds_of(ntt, gpu);
        const int npgi = npg_inst(gpu, (int)((Wn + 63) / 64));
        const dim3 grid(ua2_ceil_div(ntiles, tpw), p_rb, a->B);
        const int cpu = gpu * a->K;
#define UA2_X3P(N, C, G, P) launch_x3p<N, C, G, P>(*a, grid, smem, p_rt, tpw, ntiles, ngroups == gpu, st)
#define UA2_X3P_N(C, G, P) (ntt == 4 ? UA2_X3P(4, C, G, P) : ntt == 2 ? UA2_X3P(2, C, G, P) : UA2_X3P(1, C, G, P))
        if (nw == 8) {
          UA2_CHECK(ntt == 4 && cpu == 7 && npgi == 2, "ua2_conv1d: fused 128-channel unit outside its instantiation (ntt=%d K=%d npg=%d)", ntt, a->K, npgi);
          launch_x3p<4, 7, 1, 2, 8>(*a, grid, smem, p_rt, tpw, ntiles, false, st);
        } else
        switch ((cpu * 8 + gpu) * 8 + npgi) {
          case (7 * 8 + 1) * 8 + 1: UA2_X3P_N(7, 1, 1); break;
          case (7 * 8 + 1) * 8 + 2: UA2_X3P_N(7, 1, 2); break;
          case (7 * 8 + 1) * 8 + 3: UA2_X3P_N(7, 1, 3); break;
          case (1 * 8 + 1) * 8 + 2: UA2_X3P_N(1, 1, 2); break;
          case (1 * 8 + 1) * 8 + 4: UA2_X3P_N(1, 1, 4); break;
          case (2 * 8 + 1) * 8 + 2: UA2_X3P_N(2, 1, 2); break;
          case (2 * 8 + 1) * 8 + 4: UA2_X3P_N(2, 1, 4); break;
          case (2 * 8 + 2) * 8 + 1: UA2_X3P_N(2, 2, 1); break;
          case (2 * 8 + 2) * 8 + 2: UA2_X3P_N(2, 2, 2); break;
          case (4 * 8 + 2) * 8 + 1: UA2_X3P_N(4, 2, 1); break;
          case (4 * 8 + 2) * 8 + 2: UA2_X3P_N(4, 2, 2); break;
          case (4 * 8 + 4) * 8 + 1: UA2_X3P_N(4, 4, 1); break;
          case (8 * 8 + 4) * 8 + 1: UA2_X3P_N(8, 4, 1); break;
          default: UA2_CHECK(false, "ua2_conv1d: no pipelined instantiation for K=%d gpu=%d npg=%d", a->K, gpu, npgi);
        }
#undef UA2_X3P_N
#undef UA2_X3P
        UA2_LAUNCH_CHECK();
        return 0;
      }
    }
    int ntt = 4;
    while (ntt > 1 && (int64_t)ua2_ceil_div(tq, 16 * ntt * (4 / rt)) * row_blocks3 * a->B < 512) ntt >>= 1;
    auto lds_bytes = [&](int n) {
      const int wg = 16 * n * (4 / rt);
      return (size_t)2 * ((wg - 1) * a->stride + (a->K - 1) * a->dilation + 1) * kRowB + (a->w2 ? (size_t)(a->Cout / 32) * 2 * wg * kRowB : 0);
    };
    while (ntt > 1 && lds_bytes(ntt) > 52 * 1024) ntt >>= 1;            // keep three workgroups per CU resident (latency hiding beats tile size here)
    const int wgt = 16 * ntt * (4 / rt);
    const int W3 = (wgt - 1) * a->stride + (a->K - 1) * a->dilation + 1;
    const size_t smem3 = lds_bytes(ntt);
    UA2_CHECK(smem3 <= 150 * 1024, "ua2_conv1d: window too large (%zu B LDS)", smem3);
    const dim3 grid3(ua2_ceil_div(tq, wgt), row_blocks3, a->B);
    if (rpw == 2) {
      if (ntt == 4) launch_x3<4, 2>(*a, grid3, smem3, rt, st);
      else if (ntt == 2) launch_x3<2, 2>(*a, grid3, smem3, rt, st);
      else launch_x3<1, 2>(*a, grid3, smem3, rt, st);
    } else {
      if (ntt == 4) launch_x3<4, 1>(*a, grid3, smem3, rt, st);
      else if (ntt == 2) launch_x3<2, 1>(*a, grid3, smem3, rt, st);
      else launch_x3<1, 1>(*a, grid3, smem3, rt, st);
    }
    UA2_LAUNCH_CHECK();
    return 0;
  }
  UA2_CHECK(a->precision == 0, "ua2_conv1d: precision must be 0 (exact fp32) or 1 (bf16 x 3)");
  const int row_blocks = ua2_ceil_div((int64_t)a->Cout * a->out_phases, kBR);
  // largest time tile that still gives >= 512 workgroups (2 per CU); never below 16 steps
  int ntt = 4;
  static const int min_wg = getenv("UA2_CONV_MIN_WG") ? atoi(getenv("UA2_CONV_MIN_WG")) : 512;   // experiment hook
  while (ntt > 1 && (int64_t)ua2_ceil_div(tq, 16 * ntt) * row_blocks * a->B < min_wg) ntt >>= 1;
  const int bt = 16 * ntt;
  const int W = (bt - 1) * a->stride + (a->K - 1) * a->dilation + 1;
  const size_t smem = (size_t)kCIG * (W + 1) * sizeof(float) + (size_t)kCIG * a->K * sizeof(int);
  UA2_CHECK(smem <= 150 * 1024, "ua2_conv1d: window too large (%zu B LDS)", smem);
  ua2_allow_big_lds<conv1d_kernel<4>>();
  ua2_allow_big_lds<conv1d_kernel<2>>();
  ua2_allow_big_lds<conv1d_kernel<1>>();
  const dim3 grid(ua2_ceil_div(tq, bt), row_blocks, a->B);
  if (ntt == 4) hipLaunchKernelGGL(conv1d_kernel<4>, grid, dim3(256), smem, (hipStream_t)stream, *a);
  else if (ntt == 2) hipLaunchKernelGGL(conv1d_kernel<2>, grid, dim3(256), smem, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(conv1d_kernel<1>, grid, dim3(256), smem, (hipStream_t)stream, *a);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_avgpool1d(const float* x, float* y, int64_t rows, int32_t Tin, int32_t k, void* stream) {
  UA2_CHECK(x && y && rows > 0 && Tin > 0 && k > 0 && Tin >= k, "ua2_avgpool1d: bad arguments");
  const int Tout = Tin / k;
  const int64_t total = rows * Tout;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(avgpool1d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, rows, Tin, Tout, k);
  UA2_LAUNCH_CHECK();
  return 0;
}
