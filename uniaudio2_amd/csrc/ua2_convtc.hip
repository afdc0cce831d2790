// Decode-side 1-D convolutions of the codec on time-major split planes (ua2_conv1d_tc; include/ua2hip.h has the contract).
//
// Replaces, for ScalarModel.decode (tools/tokenizer/ReasoningCodec_film/models/scalar24k.py:403-407): Conv1d :36-74,
// ConvTranspose1d :76-112 as phase filters, ResidualUnit :143-151 (fused: conv k7 -> PReLU -> 1 x 1 conv -> PReLU -> + x in one
// launch), the repeat-upsampling of PostProcessor :136-140 and the PReLU epilogues.  Same arithmetic as ua2_conv1d's
// precision-1 (bf16 x 3) form; what changes is how activations travel: hi / lo bf16 planes [B][T][C] instead of fp32
// [B][C][T].
//
// Why (cycle stamps of the round-3 kernel, profiles/r3_notes.md §8): with fp32 [C][T] activations a fused 64-channel unit
// spent 30 % of its time requesting the window into registers, converting it to hi / lo bf16 and transposing it into the
// time-major LDS image the MFMA B fragments are read from, and the registers that held the window in flight left none for a
// fragment read-ahead set (the MFMA loop ran at 1/3 of its issue rate: read -> wait -> multiply per chunk).  In this layout
// an MFMA B fragment (8 consecutive channels of one time step) is 16 contiguous bytes in HBM, so
//   * the window goes L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no VALU
//     conversion, no transposing ds_write); zero padding = lanes pointed at a 16-byte zero source; the bank swizzle of the
//     image is applied by choosing WHICH 16 bytes each lane fetches (an LDS-DMA writes lane i at base + 16 i);
//   * the freed registers hold a second fragment set: chunk c + 1's fragments are read while chunk c is multiplied;
//   * a wave owns two row tiles (32 output channels): every B fragment read feeds 6 MFMAs instead of 3 — half the LDS read
//     traffic per flop of the round-3 kernel, which was within 1.5x of the LDS peak;
//   * the residual of a fused unit is taken from the LDS window (x is read from HBM once per unit), the epilogue writes the
//     next layer's operand (hi / lo split once, by the producer).
// Waits on the LDS-DMA are hand-counted (the compiler does not see that a DMA feeds the ds_reads behind it): see `unit_end`.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "ua2_common.h"

// Phase-knockout experiments (profiles/r4_notes.md): -DUA2_TC_DBG=<bits> builds a TIMING-ONLY library (results are wrong):
// 1 no weight refills, 2 no window DMA after the prologue, 4 no MFMAs in the chunk loop, 8 no output stores, 16 no fragment reads;
// big-tile kernel: 256 no window requests at all (LDS holds whatever it held), 512 no output stores — together: what a unit of a FUSED
// stage would cost with its input and output staying on the chip (profiles/r6_notes.md §15)
#ifndef UA2_TC_DBG
#define UA2_TC_DBG 0
#endif

#if UA2_TC_DBG & 32
// cycle stamps (bit 32): wave 0 of workgroup 0, tile index 1 of its run; read back through ua2_tc_stamps (debug builds only)
__device__ unsigned long long g_tc_stamp[64];
#define UA2_STAMP(slot)                                                                                        \
  do {                                                                                                         \
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp_on) g_tc_stamp[(slot)] = __builtin_readcyclecounter();   \
  } while (0)
extern "C" int ua2_tc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tc_stamp), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#else
#define UA2_STAMP(slot) do { } while (0)
#endif

namespace {

constexpr int kG = 32;        // channels per group = K extent of one MFMA chunk
constexpr int kRowP = 80;     // plain kernel: LDS bytes per window position per plane (64 B + 16 B pad)

__device__ __attribute__((aligned(16))) unsigned g_tc_zero[4];   // zero source of the padding rows (LDS-DMA reads it)

// LDS-DMA of 16 bytes per lane: lane i's bytes land at LDS address `lds_addr` + 16 i.  Spelled as inline asm on purpose: through
// the builtin the compiler books the instruction as a FLAT access that may touch LDS *and* memory, and while one is pending it
// resolves EVERY vmcnt / lgkmcnt dependency with a full drain (s_waitcnt vmcnt(0) lgkmcnt(0) in front of the first MFMA of
// every unit: the weight refills it was supposed to count).  Hidden from its bookkeeping, the compiler's own waits stay exact
// for the loads it knows and can only be stricter than needed (a hidden operation is one more that has to retire first).
__device__ __forceinline__ void lds_dma16(const void* gptr, unsigned lds_addr) {
  // M0 is compiler-reserved: a clobber entry for it is ignored (with a warning per call site), so it is not listed.  What makes the
  // overwrite safe is that hipcc keeps nothing in M0 in these kernels (gfx9 DS instructions do not use it) — a property of the
  // compiled code, asserted there: tests/test_isa_waits.py::test_compiler_never_touches_m0_around_the_dma.
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory");
}
// The same with the address split as  wave-uniform base (SGPR pair) + per-lane 32-bit offset: the request of an INTERIOR window —
// every row inside the sequence, no repeat-upsampling — costs a handful of SALU instructions instead of ~35 VALU ones (clamps, the
// zero-source select, a 64-bit multiply-add per lane: measured 950-1270 cycles for four requests, profiles/r4_notes.md §3; a unit of
// the k = 1 convs issues eight per wave against 48 MFMAs).  `s_nop 4`: the base is a fresh SALU result and nothing inside an asm
// string is padded by the compiler (SALU write -> VMEM read of the SGPR: 5 wait states).
__device__ __forceinline__ void lds_dma16_s(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}

__device__ __forceinline__ float prelu1(float v, float a) { return v >= 0.f ? v : __fmul_rn(a, v); }

// 4 consecutive channels from their hi / lo words: x = hi + lo (exact: 16 significant bits)
__device__ __forceinline__ void join4(uint2 h, uint2 l, float (&x)[4]) {
  x[0] = __fadd_rn(__uint_as_float(h.x << 16), __uint_as_float(l.x << 16));
  x[1] = __fadd_rn(__uint_as_float(h.x & 0xffff0000u), __uint_as_float(l.x & 0xffff0000u));
  x[2] = __fadd_rn(__uint_as_float(h.y << 16), __uint_as_float(l.y << 16));
  x[3] = __fadd_rn(__uint_as_float(h.y & 0xffff0000u), __uint_as_float(l.y & 0xffff0000u));
}
__device__ __forceinline__ void split4(const float (&v)[4], uint2& h, uint2& l) {
  split_pair(v[0], v[1], h.x, l.x);
  split_pair(v[2], v[3], h.y, l.y);
}

__global__ void tc_pack_kernel(const float* __restrict__ x, unsigned* __restrict__ hi, unsigned* __restrict__ lo, int B, int C, int T) {
  const int64_t total = (int64_t)B * T * (C / 2);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c2 = (int)(idx % (C / 2));
    const int64_t bt = idx / (C / 2);
    const int t = (int)(bt % T), b = (int)(bt / T);
    const float x0 = x[((size_t)b * C + 2 * c2) * T + t], x1 = x[((size_t)b * C + 2 * c2 + 1) * T + t];
    unsigned h, l;
    split_pair(x0, x1, h, l);
    hi[idx] = h;
    lo[idx] = l;
  }
}

__global__ void tc_unpack_kernel(const unsigned short* __restrict__ hi, const unsigned short* __restrict__ lo, float* __restrict__ y,
                                 int B, int C, int T) {
  const int64_t total = (int64_t)B * C * T;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx % T);
    const int64_t bc = idx / T;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const size_t o = ((size_t)b * T + t) * C + c;
    y[idx] = __fadd_rn(bf2f(hi[o]), bf2f(lo[o]));
  }
}

// Store of one lane's 4 consecutive output rows n0 .. n0 + 3 (one time step): planes or fp32 [C][T].
// `v` already carries bias / activation / residual.
template <int F32 = -1>   // -1: decided at run time (plain kernel); 0 / 1: the pipelined instantiations know
__device__ __forceinline__ void tc_store4(const ua2_convtc_args& a, int b, int n0, int rows, int t, const float (&v)[4]) {
  if (n0 >= rows) return;
  const int phase = n0 / a.Cout, co = n0 - phase * a.Cout;
  const int to = t * a.out_phases + phase - a.out_trim_left;
  if (to < 0 || to >= a.Tout) return;
  if ((UA2_TC_DBG & 8) && F32 >= 0 && v[0] != 123.456f) return;
  if (F32 == 1 || (F32 < 0 && a.y_f32)) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < rows) a.y_f32[((size_t)b * a.Cout + co + r) * a.Tout + to] = v[r];   // Cout < 4 only for the 1-channel waveform: rows stay in phase 0
  } else {
    uint2 h, l;
    split4(v, h, l);
    const size_t o = ((size_t)b * a.Tout + to) * a.Cout + co;
    *reinterpret_cast<uint2*>(a.y_hi + o) = h;
    *reinterpret_cast<uint2*>(a.y_lo + o) = l;
  }
}

// ---- plain form: phase by phase, any K; the reference of the bit-identity test and the fallback for unusual shapes ----
template <int NTT, int RPW>
__global__ __launch_bounds__(256) void convtc_plain_kernel(const ua2_convtc_args a, const int rt) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const int K = a.K, d = a.dilation;
  const int tsub = 4 / rt;
  constexpr int kBT = 16 * NTT;
  const int wgt = kBT * tsub;
  const int W = wgt + (K - 1) * d;
  char* xh = smc;
  char* xl = smc + (size_t)W * kRowP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tl = lane & 15, g = lane >> 4;
  const int wr = wave % rt, wt = wave / rt;
  const int t0 = blockIdx.x * wgt, tw0 = wt * kBT;
  const int r0 = (blockIdx.y * rt + wr) * (16 * RPW);
  const int b = blockIdx.z;
  const int rows = a.Cout * a.out_phases;
  const int ngroups = a.Cin / kG, nchunks = ngroups * K;
  const int tin_eff = a.Tin * a.in_repeat;
  const bool wave_active = r0 < rows;
  const int ntile_rows = (rows + 15) / 16;
  const u32x4* wph[RPW];
  const u32x4* wpl[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int tile = min(r0 / 16 + q, ntile_rows - 1);
    wph[q] = reinterpret_cast<const u32x4*>(a.w) + (size_t)tile * nchunks * 64 + lane;
    wpl[q] = reinterpret_cast<const u32x4*>(a.w_lo) + (size_t)tile * nchunks * 64 + lane;
  }
  const int in_start = t0 - a.pad_left;
  const unsigned* xhi32 = reinterpret_cast<const unsigned*>(a.x_hi) + (size_t)b * a.Tin * (a.Cin / 2);
  const unsigned* xlo32 = reinterpret_cast<const unsigned*>(a.x_lo) + (size_t)b * a.Tin * (a.Cin / 2);

  f32x4 acc[RPW][NTT];
#pragma unroll
  for (int q = 0; q < RPW; ++q)
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int cg = 0; cg < ngroups; ++cg) {
    __syncthreads();
    for (int idx = tid; idx < W * 16; idx += 256) {
      const int wi = idx >> 4, p = idx & 15;
      const int ti = in_start + wi;
      unsigned h = 0u, l = 0u;
      if (ti >= 0 && ti < tin_eff) {
        const size_t off = (size_t)(ti / a.in_repeat) * (a.Cin / 2) + cg * 16 + p;
        h = xhi32[off];
        l = xlo32[off];
      }
      *reinterpret_cast<unsigned*>(xh + (size_t)wi * kRowP + p * 4) = h;
      *reinterpret_cast<unsigned*>(xl + (size_t)wi * kRowP + p * 4) = l;
    }
    __syncthreads();
    if (wave_active) {
      for (int j = 0; j < K; ++j) {
        const int chunk = cg * K + j;
        bf16x8 ah[RPW], al[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          ah[q] = __builtin_bit_cast(bf16x8, wph[q][(size_t)chunk * 64]);
          al[q] = __builtin_bit_cast(bf16x8, wpl[q][(size_t)chunk * 64]);
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const size_t o = (size_t)(tw0 + nt * 16 + tl + j * d) * kRowP + g * 16;
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
    // fused residual unit: the workgroup holds all C output channels of its time tile (launcher: gridDim.y == 1)
    const int C = a.Cout, ng2 = C / kG;
    char* hbase = smc + 2 * (size_t)W * kRowP;                          // [ng2][2 planes][wgt][kRowP]
    const size_t hplane = (size_t)wgt * kRowP;
    __syncthreads();
    if (wave_active) {
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int nb = r0 + q * 16 + g * 4;
        if (nb >= rows) continue;
        const int grp = nb / kG, pc = nb % kG;
        float bs[4], al1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bs[r] = a.bias ? a.bias[nb + r] : 0.f;
          al1[r] = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? nb + r : 0] : 1.f;
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          float hv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = prelu1(__fadd_rn(acc[q][nt][r], bs[r]), al1[r]);
          uint2 hh, hl;
          split4(hv, hh, hl);
          // position of channel pc .. pc + 3 inside the group's K extent: k' = 8 g + 4 (row tile & 1) + r (the fused units' K order,
          // see tc_w2_order below): the lane's 4 values of an even row tile sit in front of its 4 values of the odd one
          const int kp = ((pc & 12) << 1) | ((pc & 16) >> 2);
          char* dst = hbase + (size_t)grp * 2 * hplane + (size_t)(tw0 + nt * 16 + tl) * kRowP + kp * 2;
          *reinterpret_cast<uint2*>(dst) = hh;
          *reinterpret_cast<uint2*>(dst + hplane) = hl;
        }
      }
    }
    __syncthreads();
    if (!wave_active) return;
    f32x4 acc2[RPW][NTT];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) acc2[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cg = 0; cg < ng2; ++cg) {
      bf16x8 ah[RPW], al[RPW];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const size_t wo = ((size_t)min(r0 / 16 + q, ntile_rows - 1) * ng2 + cg) * 64 + lane;
        ah[q] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(a.w2)[wo]);
        al[q] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(a.w2_lo)[wo]);
      }
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const size_t o = (size_t)cg * 2 * hplane + (size_t)(tw0 + nt * 16 + tl) * kRowP + g * 16;
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
    for (int q = 0; q < RPW; ++q) {
      const int n0 = r0 + q * 16 + g * 4;
      if (n0 >= rows) continue;
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const int t = t0 + tw0 + nt * 16 + tl;
        if (t >= a.Tout) continue;
        const size_t xo = ((size_t)b * a.Tin + t) * a.Cin + n0;                        // the residual is the unit's input
        float xr[4], v[4];
        join4(*reinterpret_cast<const uint2*>(a.x_hi + xo), *reinterpret_cast<const uint2*>(a.x_lo + xo), xr);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __fadd_rn(prelu1(__fadd_rn(acc2[q][nt][r], a.bias2 ? a.bias2[n0 + r] : 0.f), alpha2), xr[r]);
        tc_store4(a, b, n0, rows, t, v);
      }
    }
    return;
  }
  if (!wave_active) return;
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int n0 = r0 + q * 16 + g * 4;
    if (n0 >= rows) continue;
    float bs[4], al1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = (n0 + r) % a.Cout;
      bs[r] = (a.bias && n0 + r < rows) ? a.bias[co] : 0.f;
      al1[r] = (a.post_act == UA2_ACT_PRELU && n0 + r < rows) ? a.post_alpha[a.post_alpha_n > 1 ? co : 0] : 1.f;
    }
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      const int t = t0 + tw0 + nt * 16 + tl;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = prelu1(__fadd_rn(acc[q][nt][r], bs[r]), al1[r]);
      if (a.res_hi) {
        const int phase = n0 / a.Cout, co = n0 - phase * a.Cout;
        const int to = t * a.out_phases + phase - a.out_trim_left;
        if (to >= 0 && to < a.Tout) {
          const size_t ro = ((size_t)b * a.Tout + to) * a.Cout + co;
          float xr[4];
          join4(*reinterpret_cast<const uint2*>(a.res_hi + ro), *reinterpret_cast<const uint2*>(a.res_lo + ro), xr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __fadd_rn(v[r], xr[r]);
        }
      }
      tc_store4(a, b, n0, rows, t, v);
    }
  }
}

// ---- software-pipelined LDS-DMA form ----------------------------------------------------------------------------------
// A workgroup walks `tpw` consecutive time tiles; a tile is `upt` UNITS, a unit = GPU channel groups x K taps = CPU MFMA
// chunks.  NW waves = WR row-waves x WT time-waves; a wave owns RPW row tiles (16 output rows each) x NTT time tiles of 16.
//
// LDS: two window images (unit u in image u & 1), each [2 planes][Wr rows][RB = 64 GPU bytes]; row = window position,
// 16-byte slot s of a row holds channel octet s ^ f(row), f(P) = ((P >> SH) & (2 GPU - 1)) << 1 with SH = 2 - log2(GPU): with
// that map the four 16-lane groups of a ds_read_b128 (lanes {0-3,12-15,20-27}, ... : MI355X_MICROARCH.md §LDS) touch
// every bank exactly once for any window offset (checked exhaustively: tools/ubench/tc_swizzle.py).  The fused unit adds
// an h image [2 planes][wgt][C channels] (same map over C / 8 slots) and the 1 x 1 conv's packed weights.
//
// Per unit and wave (steady state):
//   chunk loop: ds_read fragments of chunk c + 1 -> the other set | 3 x RPW x NTT MFMAs of chunk c | weights of chunk c
//               refilled in place for the NEXT unit (plain global loads: the compiler counts those)
//   unit end:   s_waitcnt vmcnt(#refills) — in-order return: everything OLDER than the refills has landed, i.e. this wave's
//               pieces of the next unit's window, requested one unit ago — lgkmcnt(0), s_barrier: everyone's pieces have, and
//               everyone is done reading this unit's image -> request the unit after next into it.
// The count in `unit_end` is the number of VMEM loads this wave issues between that request and the wait (refills; plus
// the residual loads of an un-fused conv2 at a tile's first unit); stores in between only make the wait stricter.
// K order of the fused units' 1 x 1 conv ("tc_w2_order").  Inside each 32-channel group the reduction index runs
//   k' = 8 g + e  <->  channel 4 g + e (e < 4)  |  16 + 4 g + (e - 4) (e >= 4),      g = 0 .. 3
// i.e. MFMA B-fragment lane (g, t) carries, for time step t, the 4 channels an MFMA D-fragment lane (g, t) holds of the even
// row tile followed by its 4 channels of the odd row tile.  With that order a wave that owns all output rows feeds the 1 x 1
// conv straight from its accumulator registers (convtc_big_kernel: no h image, no barrier); the kernels that exchange h
// through LDS write their image in the same order, and the host packs W2 with its input channels permuted accordingly
// (ops.tc_w2_order) — one arithmetic for all three kernels (bit-identical, tested).
template <int GPU> struct TcSw {
  static constexpr int SH = GPU == 1 ? 2 : (GPU == 2 ? 1 : 0);
  static constexpr int M = 2 * GPU - 1;
  static __device__ __forceinline__ unsigned f(unsigned P) { return ((P >> SH) & M) << 1; }
};

// MODE 0: plain epilogue, 1: separate residual planes, 2: fused unit.  HALO: largest (K - 1) * dilation the instantiation serves —
// it fixes the number of LDS-DMA instructions a wave issues per unit at compile time (surplus ones fetch the zero source into a
// dump slot): with a run-time count the compiler can no longer tell how many VMEM operations separate a weight refill from its
// first use and drains vmcnt to 0 at the top of every unit (seen in the first version's ISA).
template <int NTT, int RPW, int CPU, int GPU, int NW, int WR, int MODE, int HALO>
__global__ __launch_bounds__(64 * NW, 2) void convtc_pipe_kernel(const ua2_convtc_args a, const int tpw, const int ntiles, const int nrb,
                                                                 const int ngx) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  constexpr int K = CPU / GPU;
  constexpr int RB = 64 * GPU;                           // image row bytes per plane
  constexpr int RPI = 1024 / RB;                         // rows per DMA instruction
  constexpr int SPR = 4 * GPU;                           // 16-byte slots per row
  constexpr bool FUSED = MODE == 2, RES = MODE == 1;
  constexpr bool F32OUT = RPW == 1 && WR == 1;           // the one-row-tile instantiation (waveform conv) writes fp32 [C][T], all others planes
  constexpr int kBT = 16 * NTT;
  constexpr int WT = NW / WR;
  constexpr int wgt = kBT * WT;
  constexpr int NDMA_MAX = 2 * ((wgt + HALO + RPI - 1) / RPI);
  constexpr int DPW = (NDMA_MAX + NW - 1) / NW;          // LDS-DMA instructions per wave and unit
  const int d = a.dilation;
  const int W = wgt + (K - 1) * d;                       // launcher: (K - 1) * d <= HALO
  const int Wr = (W + RPI - 1) / RPI * RPI;
  const unsigned planeB = (unsigned)Wr * RB, bufB = 2 * planeB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, g = lane >> 4;
  const int wrow = wave % WR, wt = wave / WR;
  const int tw0 = wt * kBT;
  const int wg_rb = blockIdx.x % nrb, wg_rest = blockIdx.x / nrb;
  const int wg_tx = wg_rest % ngx, b = wg_rest / ngx;
  const int rtile0 = (wg_rb * WR + wrow) * RPW;          // this wave's first row tile
  const int rows = a.Cout * a.out_phases;
  const int ntile_rows = (rows + 15) / 16;
  const int ngroups = a.Cin / kG;
  const int nchunks = ngroups * K;
  const int upt = ngroups / GPU;
  const int tin_eff = a.Tin * a.in_repeat;
  const int tile_first = wg_tx * tpw;
  if (tile_first >= ntiles) return;
  const int my_tiles = min(tpw, ntiles - tile_first);
  const int n_units = my_tiles * upt;

  // weight addressing: uniform base (+ chunk * 1 KiB, scalar) + one 32-bit per-lane offset per row tile, shared by the hi and lo
  // buffers (64-bit per-lane pointers cost 8 registers here); launcher: the packed filter is < 4 GiB
  const char* wbh = reinterpret_cast<const char*>(a.w);
  const char* wbl = reinterpret_cast<const char*>(a.w_lo);
  unsigned woff[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) woff[q] = ((unsigned)min(rtile0 + q, ntile_rows - 1) * (unsigned)nchunks * 64u + (unsigned)lane) * 16u;   // a wave past the last row tile computes on a clamped tile and stores nothing
  auto wload = [&](const char* base, int q, size_t chunk) { return *reinterpret_cast<const u32x4*>(base + chunk * 1024 + woff[q]); };
  const char* xhb = reinterpret_cast<const char*>(a.x_hi) + (size_t)b * a.Tin * a.Cin * 2;
  const char* xlb = reinterpret_cast<const char*>(a.x_lo) + (size_t)b * a.Tin * a.Cin * 2;
  const unsigned rep_magic = a.in_repeat > 1 ? (unsigned)(0x100000000ull / (unsigned)a.in_repeat + 1) : 0u;

  // LDS map: [window image 0][window image 1][fused: h image, W2 image][1 KiB dump slot of the surplus DMA instructions]
  const unsigned fusedB = FUSED ? (unsigned)(2 * wgt * a.Cout * 2 + 2 * a.Cout * a.Cout * 2 + 3 * a.Cout * 4) : 0u;
  const unsigned dump_off = 2 * bufB + fusedB;
  // ---- LDS-DMA of one unit's window: instruction i = (row block i >> 1, plane i & 1); a wave issues DPW of them, straight-line ----
  const int n_dma = 2 * (Wr / RPI);
  const int row_l = lane / SPR, slot_l = lane % SPR;
  // Interior windows (all Wr rows inside the sequence, input not repeat-upsampled): a request is uniform base (SGPR pair) + one
  // per-lane offset (lds_dma16_s) — a few SALU instructions instead of ~35 VALU ones (clamps, the zero-source select, a 64-bit
  // multiply-add per lane).  Worth 3 % of the decode (profiles/r6_notes.md §15: the ~250 cycles a wave spends per request are mostly
  // the LDS-DMA's own issue rate, not this arithmetic; a per-REQUEST choice that also served the edge tiles measured slower).
  // A wave's requests all go to ONE plane (i & 1 = wave & 1: NW is even) and to rows r0i + row_l with r0i advancing by RPI NW / 2,
  // a multiple of 8 — the slot map f() only looks at row bits 0 .. 2, so the lane's channel octet is the same for every request.
  constexpr bool FASTOK = NW % 2 == 0 && (RPI * NW / 2) % 8 == 0 && !(UA2_TC_DBG & 64);
  const unsigned rowB = (unsigned)a.Cin * 2u;
  const unsigned voff_l = (unsigned)row_l * rowB + ((((unsigned)slot_l) ^ TcSw<GPU>::f((unsigned)((wave >> 1) * RPI + row_l))) << 4);
  const char* const plane_w = (wave & 1) ? xlb : xhb;
  auto dma_unit = [&](int ti, int ug, unsigned buf_off) __attribute__((always_inline)) { // (tile index in this workgroup's run, unit in tile): clamped by the caller
    const int in_start = (tile_first + ti) * wgt - a.pad_left;
    const unsigned cbyte = (unsigned)(ug * GPU * kG) * 2u;
    if (FASTOK && rep_magic == 0u && in_start >= 0 && in_start + Wr <= tin_eff) {     // uniform
      const uint64_t base_v = (uint64_t)(uintptr_t)plane_w + ((uint64_t)(unsigned)in_start * rowB + cbyte);
      const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base_v >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base_v);   // uniform by construction; said so to the compiler
#pragma unroll
      for (int k = 0; k < DPW; ++k) {
        const int i = wave + k * NW;
        const bool live = i < n_dma;                      // uniform; a surplus instruction re-reads the window's first rows into the dump slot
        const int r0i = live ? (i >> 1) * RPI : (wave >> 1) * RPI;
        const unsigned ldst = live ? buf_off + (unsigned)(i & 1) * planeB + (unsigned)r0i * RB : dump_off;
        lds_dma16_s(reinterpret_cast<const char*>((uintptr_t)(base + (uint64_t)((unsigned)r0i * rowB))), voff_l,
                    __builtin_amdgcn_readfirstlane(lds_addr_of(smc + ldst)));
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
      const int i = wave + k * NW;
      const bool live = i < n_dma;                        // uniform; a surplus instruction reads the zero source into the dump slot
      const int r0i = (i >> 1) * RPI;
      const unsigned P = (unsigned)(r0i + row_l);         // image row of this lane's slot
      const int pos = in_start + (int)P;
      const bool ok = live && pos >= 0 && pos < tin_eff;
      const int cp = min(max(pos, 0), tin_eff - 1);
      const int src = rep_magic ? (int)__umulhi((unsigned)cp, rep_magic) : cp;
      const unsigned oct = (unsigned)slot_l ^ TcSw<GPU>::f(P);
      const char* pl = (i & 1) ? xlb : xhb;
      const char* gp = pl + ((size_t)src * a.Cin * 2 + cbyte + oct * 16);
      gp = ok ? gp : reinterpret_cast<const char*>(g_tc_zero);
      char* ldst = live ? smc + buf_off + (unsigned)(i & 1) * planeB + (unsigned)r0i * RB : smc + dump_off;
      lds_dma16(gp, __builtin_amdgcn_readfirstlane(lds_addr_of(ldst)));
    }
  };

  // fused unit: h image + the 1 x 1 conv's weights behind the two window images
  const int C2 = a.Cout;                                  // FUSED: Cin == Cout == all rows of the workgroup
  const int ng2 = C2 / kG;
  const unsigned hrowB = (unsigned)C2 * 2u, hplane = (unsigned)wgt * hrowB;
  const unsigned h_off = 2 * bufB, w2_off = h_off + 2 * hplane, w2_plane = (unsigned)(C2 / 16) * ng2 * 1024u;
  const unsigned cst_off = w2_off + 2 * w2_plane;         // fused: [bias][PReLU slope 1][bias2] per channel (kept out of the registers)
  const int hsh = ng2 == 1 ? 2 : (ng2 == 2 ? 1 : 0);
  const unsigned hmask = (unsigned)(2 * ng2 - 1);
  auto fh = [&](unsigned P) { return ((P >> hsh) & hmask) << 1; };

  // ---- prologue: window of unit 0, weights of unit 0, (fused: W2 image); everything landed; window of unit 1 requested ----
  u32x4 wregh[CPU][RPW], wregl[CPU][RPW];
#pragma unroll
  for (int c = 0; c < CPU; ++c)
#pragma unroll
    for (int q = 0; q < RPW; ++q) { wregh[c][q] = wload(wbh, q, c); wregl[c][q] = wload(wbl, q, c); }
  dma_unit(0, 0, 0u);
  if constexpr (FUSED) {
    const int nblk = 2 * (C2 / 16) * ng2;                 // 1 KiB blocks: hi image then lo image, packed order verbatim
    for (int i = wave; i < nblk; i += NW) {
      const int half = nblk / 2;
      const char* src = (i < half) ? reinterpret_cast<const char*>(a.w2) + (size_t)i * 1024 : reinterpret_cast<const char*>(a.w2_lo) + (size_t)(i - half) * 1024;
      lds_dma16(src + lane * 16, __builtin_amdgcn_readfirstlane(lds_addr_of(smc + w2_off + (unsigned)i * 1024u)));
    }
  }
  if constexpr (FUSED) {
    for (int c = tid; c < C2; c += 64 * NW) {
      float* cst = reinterpret_cast<float*>(smc + cst_off);
      cst[c] = a.bias ? a.bias[c] : 0.f;
      cst[C2 + c] = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? c : 0] : 1.f;
      cst[2 * C2 + c] = a.bias2 ? a.bias2[c] : 0.f;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    const int un = min(1, n_units - 1);
    dma_unit(un / upt, un % upt, bufB);
  }

  // per-chunk fragment offsets (image-relative, time tile 0, hi plane): P = tw0 + tl + j d, slot (4 cgl + g) ^ f(P)
  unsigned bofs[CPU];
#pragma unroll
  for (int c = 0; c < CPU; ++c) {
    const unsigned P = (unsigned)(tw0 + tl + (c % K) * d);
    bofs[c] = P * RB + ((((unsigned)(c / K) * 4u + (unsigned)g) ^ TcSw<GPU>::f(P)) << 4);
  }
  // epilogue constants of this lane's rows (4 consecutive rows per row tile)
  float bias[RPW][4], alpha[RPW][4];
  int n0[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    n0[q] = (rtile0 + q) * 16 + g * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = min(n0[q] + r, rows - 1);
      const int co = n % a.Cout;
      bias[q][r] = (!FUSED && a.bias) ? a.bias[co] : 0.f;
      alpha[q][r] = (!FUSED && a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? co : 0] : 1.f;
    }
  }
  const float alpha2 = (FUSED && a.alpha2) ? a.alpha2[0] : 0.f;

  f32x4 acc[RPW][NTT];
#pragma unroll
  for (int q = 0; q < RPW; ++q)
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint2 rsh[RPW][NTT], rsl[RPW][NTT];                    // residual words (MODE 1: from the residual planes; MODE 2: from the LDS window)
#pragma unroll
  for (int q = 0; q < RPW; ++q)
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) { rsh[q][nt] = make_uint2(0u, 0u); rsl[q][nt] = make_uint2(0u, 0u); }
  constexpr int NREF = CPU * RPW * 2;                    // refill loads per unit
  constexpr int NRES = RES ? RPW * NTT * 2 : 0;          // residual loads per tile (MODE 1)

  auto request_residual = [&](int tile) {                // MODE 1: clamped addresses (always NRES loads), masked at the store
    const int t = tile * wgt + tw0 + tl;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int nn = min(n0[q], rows - 4);
      const int phase = nn / a.Cout, co = nn - phase * a.Cout;
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        const int to = min(max((t + nt * 16) * a.out_phases + phase - a.out_trim_left, 0), a.Tout - 1);
        const size_t ro = ((size_t)b * a.Tout + to) * a.Cout + co;
        rsh[q][nt] = *reinterpret_cast<const uint2*>(a.res_hi + ro);
        rsl[q][nt] = *reinterpret_cast<const uint2*>(a.res_lo + ro);
      }
    }
  };

  u32x4 fr[2][NTT][2];
  auto read_frags = [&](u32x4 (&f)[NTT][2], unsigned xb_off, int c) {
    if constexpr (UA2_TC_DBG & 16) {
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) { f[nt][0] = u32x4{bofs[c], 1u, 2u, 3u}; f[nt][1] = f[nt][0]; }
      return;
    }
    const unsigned ah = xb_off + bofs[c];
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      f[nt][0] = *reinterpret_cast<const u32x4*>(smc + ah + nt * 16 * RB);
      f[nt][1] = *reinterpret_cast<const u32x4*>(smc + ah + planeB + nt * 16 * RB);
    }
  };

  int u = 0;
  // one unit: `first` = first unit of its tile (MODE 1 counts the tile's residual loads, issued behind the standing request)
  auto unit = [&](auto first_tag, int ug) {
    constexpr bool first = decltype(first_tag)::value;
    const unsigned xb_off = (unsigned)(u & 1) * bufB;
    const size_t chunk_n = (size_t)(min(u + 1, n_units - 1) % upt) * CPU;      // refill target: the next unit's chunks (the last unit refills itself)
    [[maybe_unused]] const bool stamp_on = u / upt == ((UA2_TC_DBG & 128) ? 0 : 1);      // bit 128: the FIRST tile (launches of one tile per workgroup)
    [[maybe_unused]] const int sb = (u % upt) * 8;
    UA2_STAMP(sb + 0);
    read_frags(fr[0], xb_off, 0);
#pragma unroll
    for (int c = 0; c < CPU; ++c) {
      if (c + 1 < CPU) read_frags(fr[(c + 1) & 1], xb_off, c + 1);
      __builtin_amdgcn_sched_barrier(0);                 // the read-ahead stays AHEAD: left alone the scheduler sinks each read next to its MFMA
      bf16x8 bh[NTT], bl[NTT];
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        bh[nt] = __builtin_bit_cast(bf16x8, fr[c & 1][nt][0]);
        bl[nt] = __builtin_bit_cast(bf16x8, fr[c & 1][nt][1]);
      }
      // per accumulator the order is al*bh, ah*bl, ah*bh (small terms first, as in the plain kernel); across accumulators the
      // MFMAs interleave so that back-to-back issues are independent
      if constexpr (UA2_TC_DBG & 4) {
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) {
            asm volatile("" ::"v"(bh[nt]), "v"(bl[nt]), "v"(wregl[c][q]), "v"(wregh[c][q]));
            acc[q][nt][0] += 1.f;
          }
      } else {
#pragma unroll
      for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wregl[c][q]), bh[nt], acc[q][nt], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wregh[c][q]), bl[nt], acc[q][nt], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
          acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wregh[c][q]), bh[nt], acc[q][nt], 0, 0, 0);
      }
      if constexpr (!(UA2_TC_DBG & 1)) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {                  // refill in place: first needed one unit from now
          wregh[c][q] = wload(wbh, q, chunk_n + c);
          wregl[c][q] = wload(wbl, q, chunk_n + c);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                 // ... and the refills stay behind their chunk (sunk to the unit's end they are first needed a memory round trip later)
    }
    if constexpr (FUSED) {
      // the residual of this wave's rows = the unit's input at the tile's own positions: it sits in the window image of the
      // channel group that holds those rows (RPW == 2: row tiles 2 wrow, 2 wrow + 1 = group wrow)
      if (ug == (rtile0 * 16) / kG) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          const unsigned cin_g = (unsigned)((rtile0 + q) * 16 + g * 4) % kG;      // channel within the group
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) {
            const unsigned P = (unsigned)(a.pad_left + tw0 + nt * 16 + tl);
            const unsigned o = xb_off + P * RB + ((((cin_g >> 3)) ^ TcSw<GPU>::f(P)) << 4) + (cin_g & 4u) * 2u;
            rsh[q][nt] = *reinterpret_cast<const uint2*>(smc + o);
            rsl[q][nt] = *reinterpret_cast<const uint2*>(smc + o + planeB);
          }
        }
      }
    }
    // unit end (see the header of this kernel)
    UA2_STAMP(sb + 1);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(((UA2_TC_DBG & 1) ? 0 : NREF) + (first ? NRES : 0)) : "memory");
    UA2_STAMP(sb + 2);
    asm volatile("s_barrier" ::: "memory");
    UA2_STAMP(sb + 3);
    if constexpr (!(UA2_TC_DBG & 2)) {
      const int un = min(u + 2, n_units - 1);
      dma_unit(un / upt, un % upt, xb_off);
    }
    UA2_STAMP(sb + 4);
    asm volatile("" ::: "memory");
    ++u;
  };

  auto epilogue = [&](int tile) {
    const int t = tile * wgt + tw0 + tl;
    [[maybe_unused]] const bool stamp_on = tile == tile_first + ((UA2_TC_DBG & 128) ? 0 : 1);
    UA2_STAMP(40);
    if constexpr (FUSED) {
      // h = PReLU(conv + b1) -> hi / lo -> h image [plane][position][C], slot map fh
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const unsigned cch = (unsigned)((rtile0 + q) * 16 + g * 4);
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(smc + cst_off + cch * 4u);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(smc + cst_off + (unsigned)(C2 + (int)cch) * 4u);
          float hv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = prelu1(__fadd_rn(acc[q][nt][r], b1[r]), a1[r]);
          uint2 hh, hl;
          split4(hv, hh, hl);
          const unsigned P = (unsigned)(tw0 + nt * 16 + tl);
          const unsigned o = h_off + P * hrowB + (((((cch >> 5) << 2) | ((cch >> 2) & 3u)) ^ fh(P)) << 4) + ((cch >> 4) & 1u) * 8u;   // K order of the fused units: tc_w2_order
          *reinterpret_cast<uint2*>(smc + o) = hh;
          *reinterpret_cast<uint2*>(smc + o + hplane) = hl;
          acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      UA2_STAMP(41);
      ua2_lds_barrier();
      UA2_STAMP(42);
      f32x4 res[RPW][NTT];
#pragma unroll
      for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) res[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        if (cg >= ng2) break;
        bf16x8 ah[RPW], al[RPW], bh[NTT], bl[NTT];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          const unsigned wo = w2_off + (unsigned)(((rtile0 + q) * ng2 + cg) * 64 + lane) * 16u;
          ah[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smc + wo));
          al[q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smc + wo + w2_plane));
        }
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          const unsigned P = (unsigned)(tw0 + nt * 16 + tl);
          const unsigned o = h_off + P * hrowB + ((((unsigned)cg * 4u + (unsigned)g) ^ fh(P)) << 4);
          bh[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smc + o));
          bl[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(smc + o + hplane));
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) res[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[q], bh[nt], res[q][nt], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) res[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bl[nt], res[q][nt], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
          for (int nt = 0; nt < NTT; ++nt) res[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[q], bh[nt], res[q][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);               // one group's fragments at a time (hoisted, the four groups' A / B sets cost 128 registers)
      }
      UA2_STAMP(43);
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(smc + cst_off + (unsigned)(2 * C2 + (rtile0 + q) * 16 + g * 4) * 4u);
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          float xr[4], v[4];
          join4(rsh[q][nt], rsl[q][nt], xr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __fadd_rn(prelu1(__fadd_rn(res[q][nt][r], b2[r]), alpha2), xr[r]);
          if (t + nt * 16 < a.Tout) tc_store4<F32OUT ? 1 : 0>(a, b, n0[q], rows, t + nt * 16, v);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = prelu1(__fadd_rn(acc[q][nt][r], bias[q][r]), alpha[q][r]);
          if constexpr (RES) {
            float xr[4];
            join4(rsh[q][nt], rsl[q][nt], xr);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __fadd_rn(v[r], xr[r]);
          }
          tc_store4<F32OUT ? 1 : 0>(a, b, n0[q], rows, t + nt * 16, v);
          acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    const int tile = tile_first + ti;
    if constexpr (RES) request_residual(tile);
    unit(std::true_type{}, 0);
    for (int ug = 1; ug < upt; ++ug) unit(std::false_type{}, ug);
    epilogue(tile);
    {
      [[maybe_unused]] const bool stamp_on = ti == ((UA2_TC_DBG & 128) ? 0 : 1);
      UA2_STAMP(44);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped tail requests: nothing may still be writing LDS when the workgroup ends
}


// ---- "one big tile per workgroup" form for the narrow, long layers (C <= 64 channels at 120 / 240 / 480 kHz) -----------------
// What the phase knock-outs and cycle stamps of the kernel above say (profiles/r4_notes.md): its fused units are bound by the
// bytes a CU can take in — ~30 B/clk, the ceiling every kernel of this library has hit — because every wave re-streams its
// 28 KiB of filter per 84 MFMAs: 341 B of weights per MFMA against the ~120 B/MFMA that ceiling allows at full matrix rate.
// Weight bytes per MFMA fall with the number of time tiles a wave multiplies each fragment with, LDS bytes per MFMA with the
// number of row tiles; LDS capacity (halo + double buffering + h image) is what kept the tiles of the kernel above small.  Here:
//   * a wave owns ALL output rows (RPW = C / 16 row tiles) x NTT time tiles with RPW * NTT = 16: 64 accumulator registers,
//     48 MFMAs per filter chunk -> 85 (C = 32) / 171 (C = 64) weight bytes per MFMA, 85-170 LDS bytes per MFMA;
//   * the filter streams through a 2-chunk register ring (requested two chunks = 1500+ cycles ahead; compiler-counted), the
//     B fragments through a ring of single time tiles;
//   * a workgroup takes ONE tile of 16 * NTT * NW steps (8 waves: 512 / 1024 steps): the windows of all channel groups of the
//     tile are resident (<= 144 KiB, one workgroup per CU, two waves per SIMD), so there is no double buffering, the halo is
//     6 % instead of 85 %, group 1's window is requested in slices behind group 0's chunks, and the residual is read from the
//     window at the end;
//   * the fused unit's 1 x 1 conv takes its B operand from the accumulator registers (tc_w2_order): no h image, no barrier.
// Same products, same order per accumulator, same epilogue operations as the plain kernel: bit-identical.
// CL: the per-row constants (bias, PReLU slope, bias2) sit in LDS behind the windows (when those leave room); otherwise they are
// read from memory where they are used.
template <int RPW, int NTT, int NW, int NG, int MODE, int HALO, bool CL>
__global__ __launch_bounds__(64 * NW, 2) void convtc_big_kernel(const ua2_convtc_args a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  constexpr int K = 7, NCH = NG * K;
  constexpr bool FUSED = MODE == 2;
  constexpr int wgt = 16 * NTT * NW;
  constexpr int NDMA_MAX = 2 * ((wgt + HALO + 15) / 16);
  constexpr int DPW = (NDMA_MAX + NW - 1) / NW;          // window requests per wave and channel group
  constexpr int DPC = (DPW + K - 1) / K;                 // ... of the NEXT group, issued behind each chunk of the current one
  static_assert(RPW * NTT == 16 || !FUSED, "64 accumulator registers");
  const int d = a.dilation;
  const int W = wgt + (K - 1) * d;                       // launcher: (K - 1) * d <= HALO
  const int Wr = (W + 15) / 16 * 16;
  const unsigned planeB = (unsigned)Wr * 64u, groupB = 2 * planeB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % ntiles, b = blockIdx.x / ntiles;
  const int tw0 = wave * 16 * NTT;
  const int rows = a.Cout;
  const int ntile_rows = (rows + 15) / 16;
  const int tin_eff = a.Tin * a.in_repeat;
  const int in_start = tile * wgt - a.pad_left;
  const char* xhb = reinterpret_cast<const char*>(a.x_hi) + (size_t)b * a.Tin * a.Cin * 2;
  const char* xlb = reinterpret_cast<const char*>(a.x_lo) + (size_t)b * a.Tin * a.Cin * 2;
  const unsigned rep_magic = a.in_repeat > 1 ? (unsigned)(0x100000000ull / (unsigned)a.in_repeat + 1) : 0u;

  // window request: instruction i of a group = (row block i >> 1, plane i & 1), 16 rows x 64 B; this lane's slot: row lane >> 2,
  // 16-byte slot lane & 3 holding channel octet (lane & 3) ^ f(row) — f only looks at row bit 2, row blocks are 16-aligned
  const int n_dma = 2 * (Wr / 16);
  const int row_l = lane >> 2;
  const unsigned cb_l = (((unsigned)lane & 3u) ^ TcSw<1>::f((unsigned)row_l)) * 16u;
  // interior tiles (the whole window inside the sequence, no repeat-upsampling): uniform base + one per-lane offset, as in the
  // pipelined kernel (lds_dma16_s)
  const bool interior_in = !(UA2_TC_DBG & 64) && rep_magic == 0u && in_start >= 0 && in_start + Wr <= tin_eff;      // uniform
  const unsigned rowB = (unsigned)a.Cin * 2u;
  const unsigned voff_l = (unsigned)row_l * rowB + cb_l;
  auto dma_rows = [&](int gi, int i_req) __attribute__((always_inline)) {
    const int i = min(i_req, n_dma - 1);                  // a surplus request repeats the last one (same bytes, same place): LDS is full to the byte
    const int r0i = (i >> 1) * 16;
    if (interior_in) {
      const uint64_t base_v = (uint64_t)(uintptr_t)((i & 1) ? xlb : xhb) + ((uint64_t)(unsigned)(in_start + r0i) * rowB + (unsigned)gi * 64u);
      const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(base_v >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base_v);
      const char* ldst = smc + (unsigned)gi * groupB + (unsigned)(i & 1) * planeB + (unsigned)r0i * 64u;
      lds_dma16_s(reinterpret_cast<const char*>((uintptr_t)base), voff_l, __builtin_amdgcn_readfirstlane(lds_addr_of(ldst)));
      return;
    }
    const int pos = in_start + r0i + row_l;
    const bool ok = pos >= 0 && pos < tin_eff;
    const int cp = min(max(pos, 0), tin_eff - 1);
    const int src = rep_magic ? (int)__umulhi((unsigned)cp, rep_magic) : cp;
    const char* gp = ((i & 1) ? xlb : xhb) + ((size_t)src * a.Cin * 2 + (unsigned)gi * 64u + cb_l);
    gp = ok ? gp : reinterpret_cast<const char*>(g_tc_zero);
    const char* ldst = smc + (unsigned)gi * groupB + (unsigned)(i & 1) * planeB + (unsigned)r0i * 64u;
    lds_dma16(gp, __builtin_amdgcn_readfirstlane(lds_addr_of(ldst)));
  };

  // filter: uniform base + chunk * 1 KiB (scalar) + one 32-bit per-lane offset per row tile; 2-chunk ring
  const char* wbh = reinterpret_cast<const char*>(a.w);
  const char* wbl = reinterpret_cast<const char*>(a.w_lo);
  unsigned woff[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) woff[q] = ((unsigned)min(q, ntile_rows - 1) * (unsigned)NCH * 64u + (unsigned)lane) * 16u;
  auto wload = [&](const char* base, int q, int chunk) { return *reinterpret_cast<const u32x4*>(base + (size_t)chunk * 1024 + woff[q]); };
  constexpr int D = 2;
  u32x4 wrh[D][RPW], wrl[D][RPW];

  [[maybe_unused]] const bool stamp_on = true;
  UA2_STAMP(0);
  // ---- prologue: (constants -> LDS first: their wait must not sit behind the window requests) window of group 0, first
  // filter chunks; one drain, one barrier ----
  constexpr int C16 = RPW * 16;
  const unsigned cst_off = NG * groupB;
  if constexpr (CL) {
    float* cstw = reinterpret_cast<float*>(smc + cst_off);   // [bias][PReLU slope][bias2] per output row
    for (int c = tid; c < C16; c += 64 * NW) {
      const int co = min(c, rows - 1);
      cstw[c] = a.bias ? a.bias[co] : 0.f;
      cstw[C16 + c] = (a.post_act == UA2_ACT_PRELU) ? a.post_alpha[a.post_alpha_n > 1 ? co : 0] : 1.f;
      cstw[2 * C16 + c] = (FUSED && a.bias2) ? a.bias2[co] : 0.f;
    }
  }
  if constexpr (!(UA2_TC_DBG & 256))
    for (int i = wave; i < n_dma; i += NW) dma_rows(0, i);
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int q = 0; q < RPW; ++q) { wrh[c][q] = wload(wbh, q, min(c, NCH - 1)); wrl[c][q] = wload(wbl, q, min(c, NCH - 1)); }
  const float alpha2 = (FUSED && a.alpha2) ? a.alpha2[0] : 0.f;
  UA2_STAMP(1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  UA2_STAMP(2);
  asm volatile("s_barrier" ::: "memory");
  UA2_STAMP(3);

  // fragment offsets per tap (group 0, hi plane, time tile 0): row P = tw0 + tl + j d, slot g ^ f(P)
  unsigned bofs[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const unsigned P = (unsigned)(tw0 + tl + j * d);
    bofs[j] = P * 64u + ((((unsigned)g) ^ TcSw<1>::f(P)) << 4);
  }

  f32x4 acc[RPW][NTT];
#pragma unroll
  for (int q = 0; q < RPW; ++q)
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B fragments: ring of FR single time tiles, requested FR - 1 steps ahead of their MFMAs (step = (chunk, time tile))
  constexpr int FR = RPW >= 4 ? 3 : 4;
  u32x4 fh_[FR], fl_[FR];
  auto read_step = [&](int slot, int c, int nt) {
    const unsigned o = (unsigned)(c / K) * groupB + bofs[c % K] + (unsigned)nt * 1024u;
    fh_[slot] = *reinterpret_cast<const u32x4*>(smc + o);
    fl_[slot] = *reinterpret_cast<const u32x4*>(smc + o + planeB);
  };
  [[maybe_unused]] u32x4 w2h[RPW][NG], w2l[RPW][NG];     // fused unit: requested behind the last chunks (see `refill`)
  static_assert(!FUSED || NG <= D, "the tail of the ring fetches the 1 x 1 filter");
  constexpr int NSTEP = NCH * NTT;
#pragma unroll
  for (int s0 = 0; s0 < FR - 1; ++s0) read_step(s0 % FR, s0 / NTT, s0 % NTT);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if constexpr (NG > 1) {
      if (c == K) {
        // group boundary: this wave's slices of the next window were requested behind the chunks of group 0, the last of them
        // in FRONT of the last chunk's filter refill -> everything older than those RPW * 2 loads has landed; barrier:
        // everyone's slices have.  The fragment ring is primed again behind it (no read of the new window may pass it).
        UA2_STAMP(20);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW * 2) : "memory");
        UA2_STAMP(21);
        asm volatile("s_barrier" ::: "memory");
        UA2_STAMP(22);
#pragma unroll
        for (int s2 = K * NTT; s2 < K * NTT + FR - 1; ++s2) read_step(s2 % FR, s2 / NTT, s2 % NTT);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      const int s = c * NTT + nt;
      const int sn = s + FR - 1;
      const bool cross = NG > 1 && c < K && sn >= K * NTT;          // would read the next group's window ahead of its barrier
      if (sn < NSTEP && !cross) read_step(sn % FR, sn / NTT, sn % NTT);
      __builtin_amdgcn_sched_barrier(0);                  // the read-ahead stays ahead
      const bf16x8 bh = __builtin_bit_cast(bf16x8, fh_[s % FR]), bl = __builtin_bit_cast(bf16x8, fl_[s % FR]);
      // per accumulator: al*bh, ah*bl, ah*bh (small terms first, as in the plain kernel)
#pragma unroll
      for (int q = 0; q < RPW; ++q) acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wrl[c % D][q]), bh, acc[q][nt], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < RPW; ++q) acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wrh[c % D][q]), bl, acc[q][nt], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < RPW; ++q) acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wrh[c % D][q]), bh, acc[q][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    auto refill = [&]() {
      if (c + D < NCH) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {                  // ring refill: chunk c + D into chunk c's registers
          wrh[c % D][q] = wload(wbh, q, c + D);
          wrl[c % D][q] = wload(wbl, q, c + D);
        }
      } else if constexpr (FUSED) {
        // the ring has nothing left to fetch: its registers take the 1 x 1 conv's filter (group c + D - NCH of NG), so the
        // epilogue does not open with a memory round trip
        const int m = c + D - NCH;                        // a constant once the chunk loop is unrolled
#pragma unroll
        for (int mm = 0; mm < NG; ++mm) {
          if (mm != m) continue;
#pragma unroll
          for (int q = 0; q < RPW; ++q) {
            const size_t wo = ((size_t)(q * NG + mm) * 64 + lane) * 16;
            w2h[q][mm] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.w2) + wo);
            w2l[q][mm] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.w2_lo) + wo);
          }
        }
      }
    };
    auto slices = [&]() {
      if constexpr (NG > 1) {
        if (c < K) {
#pragma unroll
          for (int j = 0; j < DPC; ++j) {
            if constexpr (UA2_TC_DBG & 256) lds_dma16(reinterpret_cast<const char*>(g_tc_zero), __builtin_amdgcn_readfirstlane(lds_addr_of(smc)));   // keeps the hand-counted waits honest: one cheap request
            else dma_rows(1, wave + (c * DPC + j) * NW);
          }
        }
      }
    };
    // vmcnt retires in order: a window slice issued in front of a refill must land before that refill can be waited for, so
    // the slices go BEHIND the refill — except the last ones, which the hand-counted wait of the group boundary wants older
    // than exactly one refill
    if (c == K - 1) { slices(); asm volatile("" ::: "memory"); refill(); }
    else { refill(); asm volatile("" ::: "memory"); slices(); }
    __builtin_amdgcn_sched_barrier(0);
    UA2_STAMP(4 + c);
  }

  // ---- epilogue, time tile by time tile, everything in registers ----
  // per-row constants come from memory when they are used (L1 / L2 hits; in registers they would be 48 VGPRs, in LDS they
  // would cost the second workgroup per CU: the windows fill it to the byte)
  const bool chan_alpha = a.post_act == UA2_ACT_PRELU && a.post_alpha_n > 1;
  const float alpha_s = a.post_act == UA2_ACT_PRELU ? (chan_alpha ? 0.f : a.post_alpha[0]) : 1.f;
  const float* cst = reinterpret_cast<const float*>(smc + cst_off);
  auto row4 = [&](const float* p, int row, float dflt, int which) {
    if constexpr (CL) return *reinterpret_cast<const f32x4*>(cst + which * C16 + row);
    else return p ? *reinterpret_cast<const f32x4*>(p + row) : f32x4{dflt, dflt, dflt, dflt};
  };
  const int t_base = tile * wgt + tw0 + tl;
  // output: lane (g, tl) stores 4 channels (8 bytes per plane) of row tile q at step t_base + 16 nt: one base, compile-time offsets
  const size_t y0 = ((size_t)b * a.Tout + t_base) * C16 + g * 4;
  const bool interior = tile * wgt + wgt <= a.Tout && rows == C16;     // uniform: no store needs a mask
  auto store4 = [&](int q, int nt, const float (&v)[4]) {
    if ((UA2_TC_DBG & 512) && v[0] != 123.456f) return;
    if (interior) {
      uint2 h, l;
      split4(v, h, l);
      const size_t o = y0 + (size_t)nt * 16 * C16 + q * 16;
      *reinterpret_cast<uint2*>(a.y_hi + o) = h;
      *reinterpret_cast<uint2*>(a.y_lo + o) = l;
    } else if (t_base + nt * 16 < a.Tout) {
      tc_store4<0>(a, b, q * 16 + g * 4, rows, t_base + nt * 16, v);
    }
  };
  if constexpr (FUSED) {
    // residual = the unit's input at this wave's rows and steps: still in the windows.  Row P = pad_left + tw0 + 16 nt + tl (the
    // slot map only looks at bit 2 of P: the same for every nt), 8 bytes at channel (q & 1) 16 + 4 g of group q >> 1
    const unsigned P0 = (unsigned)(a.pad_left + tw0 + tl);
    const unsigned fP = TcSw<1>::f(P0);
    unsigned ro[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned cin_g = (unsigned)(e * 16 + g * 4);
      ro[e] = P0 * 64u + (((cin_g >> 3) ^ fP) << 4) + (cin_g & 4u) * 2u;
    }
    const float alpha2 = a.alpha2 ? a.alpha2[0] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NTT; ++nt) {
      UA2_STAMP(25 + nt);
      u32x4 hh[NG], hl[NG];
#pragma unroll
      for (int m = 0; m < NG; ++m) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {                     // even row tile of the group, then the odd one (tc_w2_order)
          const int row = (2 * m + e) * 16 + g * 4;
          const f32x4 b1 = row4(a.bias, row, 0.f, 0);
          const f32x4 a1 = (CL || chan_alpha) ? row4(a.post_alpha, row, 1.f, 1) : f32x4{alpha_s, alpha_s, alpha_s, alpha_s};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * e + r] = prelu1(__fadd_rn(acc[2 * m + e][nt][r], b1[r]), a1[r]);
        }
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        split_pair(v[0], v[1], h0, l0);
        split_pair(v[2], v[3], h1, l1);
        split_pair(v[4], v[5], h2, l2);
        split_pair(v[6], v[7], h3, l3);
        hh[m] = u32x4{h0, h1, h2, h3};
        hl[m] = u32x4{l0, l1, l2, l3};
      }
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        f32x4 res = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NG; ++m) {
          res = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2l[q][m]), __builtin_bit_cast(bf16x8, hh[m]), res, 0, 0, 0);
          res = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2h[q][m]), __builtin_bit_cast(bf16x8, hl[m]), res, 0, 0, 0);
          res = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2h[q][m]), __builtin_bit_cast(bf16x8, hh[m]), res, 0, 0, 0);
        }
        const unsigned o = (unsigned)(q >> 1) * groupB + ro[q & 1] + (unsigned)nt * 1024u;
        float xr[4], v[4];
        join4(*reinterpret_cast<const uint2*>(smc + o), *reinterpret_cast<const uint2*>(smc + o + planeB), xr);
        const f32x4 b2 = row4(a.bias2, q * 16 + g * 4, 0.f, 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __fadd_rn(prelu1(__fadd_rn(res[r], b2[r]), alpha2), xr[r]);
        store4(q, nt, v);
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int row = min(q * 16 + g * 4, max(rows - 4, 0));
      const f32x4 b1 = row4(a.bias, row, 0.f, 0);
      const f32x4 a1 = (CL || chan_alpha) ? row4(a.post_alpha, row, 1.f, 1) : f32x4{alpha_s, alpha_s, alpha_s, alpha_s};
#pragma unroll
      for (int nt = 0; nt < NTT; ++nt) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = prelu1(__fadd_rn(acc[q][nt][r], b1[r]), a1[r]);
        store4(q, nt, v);
      }
    }
  }
  UA2_STAMP(40);
}

template <int RPW, int NTT, int NW, int NG, int MODE, int HALO, bool CL>
int launch_big(const ua2_convtc_args& a, hipStream_t s) {
  constexpr auto kern = convtc_big_kernel<RPW, NTT, NW, NG, MODE, HALO, CL>;
  constexpr int wgt = 16 * NTT * NW;
  const int W = wgt + 6 * a.dilation, Wr = (W + 15) / 16 * 16;
  const size_t smem = (size_t)NG * 2 * Wr * 64 + (CL ? 3 * RPW * 16 * 4 : 0);   // 4 waves, 64 channels, dilation 9: windows only, 80 KiB to the byte (two workgroups per CU)
  if (smem > (NW == 8 ? 160u : 80u) * 1024u) return 1;
  const int ntiles = ua2_ceil_div(a.Tout, wgt);
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles * a.B)), dim3(64 * NW), smem, s, a, ntiles);
  UA2_LAUNCH_CHECK();
  return 0;
}

// the narrow, long layers: fused units with 32 / 64 channels, and the 32 -> 32 k7 conv behind the repeat-upsampling
int big_launch(const ua2_convtc_args& a, hipStream_t st) {
  static const bool off = getenv("UA2_CONVTC_NO_BIG") != nullptr;      // A/B hook
  if (off || a.K != 7 || a.out_phases != 1 || (a.K - 1) * a.dilation > 54 || a.res_hi || a.y_f32) return 1;
  if ((int64_t)a.B * a.Tin * a.Cin * 2 >= (1ll << 31) || (int64_t)a.Tin * a.in_repeat * a.in_repeat >= (1ll << 32)) return 1;
  if (a.Tin * a.in_repeat != a.Tout) return 1;
  // worth it only when the tiles fill the device (one 8-wave workgroup per CU).  Measured on the decoder's layers
  // (profiles/r4_notes.md): fused 64-channel unit 40.2 -> 32.1 us, 32-channel 27.1 -> 21.9 us; the un-fused 32 -> 32 conv behind
  // the repeat-upsampling 37.5 -> 39.6 us (stays on the pipelined kernel); two 4-wave workgroups per CU instead of one 8-wave
  // workgroup: 36.4 / 22.1 us.
  if (a.w2) {
    if (a.Cout == 32 && (int64_t)ua2_ceil_div(a.Tout, 1024) * a.B >= 96) return launch_big<2, 8, 8, 1, 2, 54, true>(a, st);
    if (a.Cout == 64 && (int64_t)ua2_ceil_div(a.Tout, 512) * a.B >= 96) return launch_big<4, 4, 8, 2, 2, 54, true>(a, st);
    return 1;
  }
  if (a.variant == 3 && a.Cin == 32 && a.Cout == 32 && (int64_t)ua2_ceil_div(a.Tout, 1024) * a.B >= 96) return launch_big<2, 8, 8, 1, 0, 54, true>(a, st);
  return 1;
}

template <int NTT, int RPW, int CPU, int GPU, int NW, int WR, int MODE, int HALO>
int launch_pipe(const ua2_convtc_args& a, int tpw, int ntiles, int nrb, int ngx, size_t smem, hipStream_t s) {
  constexpr auto kern = convtc_pipe_kernel<NTT, RPW, CPU, GPU, NW, WR, MODE, HALO>;
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, dim3((unsigned)(nrb * ngx * a.B)), dim3(64 * NW), smem, s, a, tpw, ntiles, nrb, ngx);
  UA2_LAUNCH_CHECK();
  return 0;
}

template <int NTT, int RPW>
int launch_plain(const ua2_convtc_args& a, dim3 grid, size_t smem, int rt, hipStream_t s) {
  constexpr auto kern = convtc_plain_kernel<NTT, RPW>;
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a, rt);
  UA2_LAUNCH_CHECK();
  return 0;
}

int plain_launch(const ua2_convtc_args& a, hipStream_t st) {
  const int rows = a.Cout * a.out_phases;
  const int tq = a.out_phases == 1 ? a.Tout : ua2_ceil_div(a.Tout + a.out_trim_left, a.out_phases);
  const int rpw = rows > 16 ? 2 : 1;
  const int wave_rows = 16 * rpw;
  const int rt = rows > 2 * wave_rows ? 4 : (rows > wave_rows ? 2 : 1);
  const int row_blocks = ua2_ceil_div(rows, wave_rows * rt);
  UA2_CHECK(!a.w2 || row_blocks == 1, "ua2_conv1d_tc: the fused unit needs all %d rows in one workgroup", rows);
  auto lds_bytes = [&](int n) {
    const int wg = 16 * n * (4 / rt);
    return (size_t)2 * (wg + (a.K - 1) * a.dilation) * kRowP + (a.w2 ? (size_t)(a.Cout / 32) * 2 * wg * kRowP : 0);
  };
  int ntt = 4;
  while (ntt > 1 && ((int64_t)ua2_ceil_div(tq, 16 * ntt * (4 / rt)) * row_blocks * a.B < 512 || lds_bytes(ntt) > 64 * 1024)) ntt >>= 1;
  const size_t smem = lds_bytes(ntt);
  UA2_CHECK(smem <= 150 * 1024, "ua2_conv1d_tc: window too large (%zu B LDS)", smem);
  const dim3 grid(ua2_ceil_div(tq, 16 * ntt * (4 / rt)), row_blocks, a.B);
  if (rpw == 2) return ntt == 4 ? launch_plain<4, 2>(a, grid, smem, rt, st) : (ntt == 2 ? launch_plain<2, 2>(a, grid, smem, rt, st) : launch_plain<1, 2>(a, grid, smem, rt, st));
  return ntt == 4 ? launch_plain<4, 1>(a, grid, smem, rt, st) : (ntt == 2 ? launch_plain<2, 1>(a, grid, smem, rt, st) : launch_plain<1, 1>(a, grid, smem, rt, st));
}

// "Row-wave" forms of the pipelined kernel (round 6): the same 64 x 64 (fused 128-channel unit: 128 x 64) workgroup tile dealt as FOUR
// (EIGHT) row-waves of one row tile x four time tiles instead of 2 x 2 waves of two row tiles x two time tiles: a wave's filter fragment
// feeds 12 MFMAs instead of 6 (170 B of filter per MFMA instead of 341, half the filter bytes a CU takes in), at twice the B-fragment
// reads from LDS.  Same chain per accumulator: same bits (tests/test_gpu_convtc.py runs both).  Measured per launch (profiles/r6_notes.md
// §15): k7 256 ch x 7500 steps 32.1 -> 28.9 us, fused 128-channel unit 39.3 -> 36.3, up-samplers 29.9 -> 26.7 / 24.9 -> 23.9,
// k1 + residual 17.3 -> 16.5; k7 512 ch x 1500 steps unchanged (31.7 / 32.0: one wave per SIMD, bound by its own serial chain).
int row_wave_forms() {
  static const int form = getenv("UA2_CONVTC_ROW_WAVES") ? atoi(getenv("UA2_CONVTC_ROW_WAVES")) : 15;   // read once; bits: 1 wide k7, 2 fused 128-channel unit, 4 up-samplers, 8 k1 + residual; 0 = the round-4 forms (A/B)
  return form;
}

// returns 1 when the shape is outside the pipelined kernel's instantiations
int pipe_launch(const ua2_convtc_args& a, hipStream_t st) {
  const int rows = a.Cout * a.out_phases;
  const int tq = a.out_phases == 1 ? a.Tout : ua2_ceil_div(a.Tout + a.out_trim_left, a.out_phases);
  const int ngroups = a.Cin / kG;
  const int mode = a.w2 ? 2 : (a.res_hi ? 1 : 0);
  if ((int64_t)a.B * a.Tin * a.Cin * 2 >= (1ll << 31) || (int64_t)a.Tin * a.in_repeat * a.in_repeat >= (1ll << 32)) return 1;
  if ((int64_t)ua2_ceil_div(rows, 16) * ngroups * a.K * 1024 >= (1ll << 32)) return 1;
  const int halo = (a.K - 1) * a.dilation;
  // (instantiation) = NTT, RPW, CPU, GPU, NW, WR, MODE, HALO
  int sel = -1, gpu = 1, ntt = 2, rpw = 2, nw = 4, wr = 2;
  if (a.K == 7 && halo <= 54) {
    if (mode == 2) {
      if (a.Cout == 32) { sel = 0; wr = 1; }
      else if (a.Cout == 64) { sel = 1; wr = 2; }
      else if (a.Cout == 128 && (row_wave_forms() & 2)) { sel = 11; ntt = 4; rpw = 1; wr = 8; nw = 8; }
      else if (a.Cout == 128) { sel = 2; wr = 4; nw = 8; }
    } else if (mode == 0) {
      if (rows <= 16) { sel = 3; ntt = 4; rpw = 1; wr = 1; }
      else if (rows == 32) { sel = 4; wr = 1; }
      else if ((row_wave_forms() & 1) && rows % 64 == 0) { sel = 10; ntt = 4; rpw = 1; wr = 4; }
      else { sel = 5; wr = 2; }
    }
  } else if (a.K == 2 && a.dilation == 1 && mode == 0 && rows >= 64) {
    if (ngroups % 4 == 0 && (row_wave_forms() & 4) && rows % 64 == 0 && (int64_t)ua2_ceil_div(tq, 64) * (rows / 64) * a.B >= 256) { sel = 13; gpu = 4; ntt = 4; rpw = 1; wr = 4; }   // fewer workgroups than CUs: the two-row-tile form (measured: 1024 -> 3 x 512 at 500 steps 26.6 vs 33-38 us)
    else if (ngroups % 4 == 0) { sel = 6; gpu = 4; }
    else if (ngroups % 2 == 0) { sel = 7; gpu = 2; }
  } else if (a.K == 1 && mode != 2 && rows >= 64 && ngroups % 4 == 0) {
    sel = mode == 1 ? 9 : 8;
    gpu = 4;
    if (mode == 1 && (row_wave_forms() & 8) && rows % 64 == 0) { sel = 12; ntt = 4; rpw = 1; wr = 4; }
  }
  if (sel < 0 || (sel == 3) != (a.y_f32 != nullptr)) return 1;    // the one-row-tile instantiation writes fp32 [C][T], the others planes
  const int wt = nw / wr;
  const int wgt = 16 * ntt * wt;
  const int rpi = 16 / gpu;
  const int64_t W = wgt + halo;
  const int64_t Wr = (W + rpi - 1) / rpi * rpi;
  size_t smem = (size_t)4 * Wr * 64 * gpu + 1024;
  if (mode == 2) smem += (size_t)2 * wgt * a.Cout * 2 + (size_t)2 * a.Cout * a.Cout * 2 + (size_t)3 * a.Cout * 4;
  if (smem > (nw == 8 ? 158u : 79u) * 1024u) return 1;
  const int ntiles = ua2_ceil_div(tq, wgt);
  const int nrb = ua2_ceil_div(rows, 16 * rpw * wr);
  const int64_t total = (int64_t)ntiles * nrb * a.B;
  const int slots = nw == 8 ? 256 : 512;
  int tpw = (int)std::min<int64_t>(8, std::max<int64_t>(1, (total + slots - 1) / slots));
  if (const char* e = getenv("UA2_CONVTC_TPW")) tpw = std::max(1, atoi(e));
  const int ngx = ua2_ceil_div(ntiles, tpw);
#define UA2_TCP(N, R, C, G, W, WRR, M, H) return launch_pipe<N, R, C, G, W, WRR, M, H>(a, tpw, ntiles, nrb, ngx, smem, st)
  switch (sel) {
    case 0: UA2_TCP(2, 2, 7, 1, 4, 1, 2, 54);             // fused unit, 32 channels: 32 rows x 128 steps
    case 1: UA2_TCP(2, 2, 7, 1, 4, 2, 2, 54);             // fused unit, 64 channels: 64 x 64
    case 2: UA2_TCP(2, 2, 7, 1, 8, 4, 2, 54);             // fused unit, 128 channels: 128 x 64, 8 waves
    case 3: UA2_TCP(4, 1, 7, 1, 4, 1, 0, 54);             // <= 16 rows (the waveform conv): 16 x 256
    case 4: UA2_TCP(2, 2, 7, 1, 4, 1, 0, 54);             // 32 rows (PostProcessor conv): 32 x 128
    case 5: UA2_TCP(2, 2, 7, 1, 4, 2, 0, 54);             // wide k7 convs: 64 x 64 per workgroup
    case 11: UA2_TCP(4, 1, 7, 1, 8, 8, 2, 54);            // fused unit, 128 channels, eight row-waves of 16 x 64
    case 12: UA2_TCP(4, 1, 4, 4, 4, 4, 1, 0);             // 1 x 1 conv with residual planes, four row-waves of 16 x 64
    case 13: UA2_TCP(4, 1, 8, 4, 4, 4, 0, 1);             // up-sampler phases, four row-waves of 16 x 64
    case 10: UA2_TCP(4, 1, 7, 1, 4, 4, 0, 54);            // ... the same tile as four row-waves of 16 x 64: half the filter bytes per MFMA (see wide_k7_form)
    case 6: UA2_TCP(2, 2, 8, 4, 4, 2, 0, 1);              // up-sampler phases (2 taps), 4 channel groups per unit
    case 7: UA2_TCP(2, 2, 4, 2, 4, 2, 0, 1);
    case 8: UA2_TCP(2, 2, 4, 4, 4, 2, 0, 0);              // 1 x 1 convs
    case 9: UA2_TCP(2, 2, 4, 4, 4, 2, 1, 0);              // ... with residual planes (second conv of a wide residual unit)
    default: return 1;
  }
#undef UA2_TCP
}

}  // namespace

extern "C" int ua2_tc_pack(const float* x, uint16_t* hi, uint16_t* lo, int32_t B, int32_t C, int32_t T, void* stream) {
  UA2_CHECK(x && hi && lo && B > 0 && C > 0 && T > 0 && C % 2 == 0, "ua2_tc_pack: bad arguments");
  const int64_t total = (int64_t)B * T * (C / 2);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(tc_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<unsigned*>(hi),
                     reinterpret_cast<unsigned*>(lo), B, C, T);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_tc_unpack(const uint16_t* hi, const uint16_t* lo, float* y, int32_t B, int32_t C, int32_t T, void* stream) {
  UA2_CHECK(y && hi && lo && B > 0 && C > 0 && T > 0, "ua2_tc_unpack: bad arguments");
  const int64_t total = (int64_t)B * T * C;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(tc_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hi, lo, y, B, C, T);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_conv1d_tc(const ua2_convtc_args* a, void* stream) {
  UA2_CHECK(a && a->x_hi && a->x_lo && a->w && a->w_lo, "ua2_conv1d_tc: NULL argument");
  UA2_CHECK(a->B > 0 && a->Cin > 0 && a->Cout > 0 && a->Tin > 0 && a->Tout > 0, "ua2_conv1d_tc: empty problem");
  UA2_CHECK(a->Cin % kG == 0, "ua2_conv1d_tc: Cin=%d must be a multiple of %d (channel groups of the matrix pipe)", a->Cin, kG);
  UA2_CHECK(a->K >= 1 && a->K <= 32 && a->dilation >= 1 && a->in_repeat >= 1 && a->out_phases >= 1 && a->pad_left >= 0 && a->out_trim_left >= 0,
            "ua2_conv1d_tc: bad geometry K=%d dil=%d", a->K, a->dilation);
  UA2_CHECK(a->out_phases == 1 || a->dilation == 1, "ua2_conv1d_tc: phase mode needs dilation 1");
  UA2_CHECK(a->post_act == UA2_ACT_NONE || (a->post_act == UA2_ACT_PRELU && a->post_alpha), "ua2_conv1d_tc: post_act is NONE or PRELU (with post_alpha)");
  UA2_CHECK((a->y_f32 != nullptr) != (a->y_hi != nullptr && a->y_lo != nullptr), "ua2_conv1d_tc: give y_hi + y_lo or y_f32");
  UA2_CHECK(a->y_f32 || a->Cout % kG == 0, "ua2_conv1d_tc: plane output needs Cout %% %d == 0", kG);
  UA2_CHECK(!a->y_f32 || a->out_phases == 1, "ua2_conv1d_tc: fp32 output has no phase mode");
  UA2_CHECK((a->res_hi == nullptr) == (a->res_lo == nullptr) && !(a->res_hi && a->w2), "ua2_conv1d_tc: residual planes come in pairs and not with a fused unit");
  UA2_CHECK(!a->res_hi || !a->y_f32, "ua2_conv1d_tc: residual planes need a plane output");
  if (a->w2) {
    UA2_CHECK(a->w2_lo && a->out_phases == 1 && a->in_repeat == 1 && a->Cin == a->Cout && (a->Cout == 32 || a->Cout == 64 || a->Cout == 128) &&
                  a->Tin == a->Tout && !a->y_f32,
              "ua2_conv1d_tc: the fused residual unit needs Cin == Cout in {32, 64, 128}, Tin == Tout, w2_lo and a plane output");
  }
  UA2_CHECK((int64_t)a->B * a->Tout * a->Cout < (1ll << 40), "ua2_conv1d_tc: problem too large");
  hipStream_t st = (hipStream_t)stream;
  if (a->variant == 0 || a->variant == 3) {
    const int rc = big_launch(*a, st);
    if (rc <= 0) return rc;
    UA2_CHECK(a->variant != 3, "ua2_conv1d_tc: shape outside the big-tile kernel's instantiations (K=%d Cin=%d Cout=%d T=%d)", a->K, a->Cin, a->Cout, a->Tout);
  }
  if (a->variant != 1) {
    const int rc = pipe_launch(*a, st);
    if (rc <= 0) return rc;
    UA2_CHECK(a->variant != 2, "ua2_conv1d_tc: shape outside the pipelined kernel's instantiations (K=%d Cin=%d Cout=%d phases=%d)", a->K, a->Cin,
              a->Cout, a->out_phases);
  }
  return plain_launch(*a, st);
}
