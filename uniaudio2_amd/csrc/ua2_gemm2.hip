// Many-row form of ua2_linear under the ORDER-FREE contract (ua2hip.h: ua2_linear_args.sum_order = UA2_SUM_ORDER_FREE).
//
// Replaces the same reference code as ua2_gemm.hip — nn.Linear call sites with the op before and after fused — for the callers that
// are NOT under the LM's row-invariance contract: the codec's flow-matching DiT (transformer_1d_flow.py:162-386, attention.py:97-420:
// q|k|v, to_out, ff.net.0 / ff.net.2), the AudioThinking encoder, the Mimi transformers, and — as a plan option — LM launches of
// >= 2048 rows (lit_model.py:424,511,591-595: prefill of batches, config 3; decode frames of > 1000 sequences).
//
// Why a second kernel.  ua2_gemm.hip reproduces the decode kernel's summation order (a row = `waves` partial MFMA chains added in
// wave order) by retiring a running chain into a second accumulator set: 2 x the accumulator registers, which caps the wave tile at
// 64 x 64 inside 256 VGPRs and the workgroup at 4 waves x 128 x 128 — the regime MI355X runs at ~0.25-0.35 of its bf16 peak, with
// the LDS-DMA issue time of a wave ADDING to its MFMA time (profiles/r4_notes.md §9).  Here a row's dot product is ONE chain over K
// in chunk order (per K slab, when a launch is cut into slabs), so:
//   * one accumulator set: a wave owns 128 x 64 (8 x 4 tiles of 16 x 16 = 128 accumulator registers),
//   * the workgroup is 8 waves = 256 x 256 (BMT = 16, one workgroup per CU) or 128 x 256 (BMT = 8, two per CU),
//   * the two waves of every SIMD run in OPPOSITE phases: while one multiplies chunk c (32 MFMAs) its partner reads
//     the fragments of its next chunk from LDS and issues its LDS-DMA requests, then they swap — the structure the guide's 256^2
//     8-phase template measures at 1.3-1.5 PFLOP/s (cdna_hip_programming.md §5; MI355X_MICROARCH.md "Two waves per SIMD").
//
// Operands: both in MFMA fragment order [tile of 16][K / 32][64 lanes][16 B] (ua2_pack_linear for the weights, the prep launch or
// the producer's y_packed for the activations) — the LDS image of a fragment block IS its global image, so an LDS-DMA request is
// one 1 KiB block (lane i's 16 bytes land at base + 16 i) and every fragment read is a conflict-free lane-linear ds_read_b128; no
// swizzle on either side.
//
// Ring: NB slots of one chunk (BMT + 16 fragment blocks).  Phases are separated by raw s_barriers; group 0 = waves 0-3 (rows
// 0 .. 8 BMT - 1 of the tile), group 1 = waves 4-7 (the other half), wave w and w + 4 share a SIMD:
//     phase        2c        2c + 1      2c + 2      2c + 3
//     group 0      L(c)      C(c)        L(c + 1)    C(c + 1)
//     group 1      C(c - 1)  L(c)        C(c)        L(c + 1)
//   L(c): read the wave's WM + 4 fragments of chunk c from slot c % NB (ds_read_b128), request the wave's LOADS blocks of chunk
//         c + NB - 1 into slot (c - 1) % NB — both groups have read chunk c - 1 by then: group 1 in phase 2c - 1 — and wait
//         lgkmcnt(0) in front of the barrier (the slot's next writer is a DMA two phases on).
//   C(c): 8 WM MFMAs.
//   landing: chunk c + 1 must be in LDS when phase 2c + 2 starts; every wave waits for ITS pieces of it — hand-counted
//         `s_waitcnt vmcnt((NB - 2) LOADS)`: the NB - 2 younger chunks stay in flight, vmcnt retires in order — behind the requests
//         of its L(c) (phase 2c for group 0, 2c + 1 for group 1), in front of a barrier every reader passes before phase 2c + 2.
// The DMA is issued through inline asm (the builtin makes hipcc drain vmcnt / lgkmcnt at every later dependency:
// profiles/r4_notes.md §2) and the waits are hand-counted; tests/test_isa_waits.py checks the compiled code for exactly that.
//
// Epilogues: the staged forms of ua2_gemm.hip — a wave parks PM x 16 rows x 64 columns of its patch (fp32) in its share of the idle
// ring and walks it by rows, 4 consecutive columns per lane — with the same operations per value, so a launch differs from
// ua2_gemm.hip's only by the order of the K sum (fp32 rounding noise, ~1e-6 relative).  STORE, RESIDUAL (+ bias, out_scale, K slabs),
// SWIGLU, GELU (+ packed hand-off), q|k|v with half-split RoPE at head size 128 (the LM) or no rotation + bias at head size 64 (the
// DiT).  Round 6: the scaled-norm hand-over on both sides (UA2_PRO_SCALED consumers read one row scale per row, reduced from the
// producer's partials by a small launch in front; y_norm_w producers emit the bf16 operand + per-16-column sums of squares from the
// RESIDUAL / STORE epilogue and from the K-split combine), the arg-max partials of STORE (lm_head / audio_head; N % 4 == 0 is
// enough: 12 296 columns), and a TAIL SPLIT for RESIDUAL launches whose last round of tiles would leave most CUs idle (300 tiles on
// 256 CUs: 256 tiles whole, the other 44 as K slabs on a second launch + combine).  Other head sizes stay on ua2_gemm.hip.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "ua2_common.h"
#include "ua2_linear_common.h"

namespace {

__device__ __forceinline__ unsigned g2_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}

// LOADS LDS-DMA requests of 1 KiB (16 B per lane) in one statement: block j comes from the wave-uniform base s_j + the lane's
// `voff` and lands at LDS address d_j + 16 lane.  M0 (the DMA's LDS base) is saved and restored inside the statement: it is
// compiler-reserved and a clobber would be ignored (cdna_hip_programming.md §5.7).  `s_nop 4`: the bases are fresh SALU results,
// and nothing inside an asm string is padded by the compiler (SALU write -> VMEM read of the SGPR).
__device__ __forceinline__ void g2_dma(const char* s0, const char* s1, const char* s2, const char* s3, unsigned voff, unsigned d0,
                                       unsigned d1, unsigned d2, unsigned d3) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %2\n\t"
      "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %3\n\t"
      "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %4\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(voff), "s"(d0), "s"(d1), "s"(d2), "s"(d3)
      : "memory");
}
__device__ __forceinline__ void g2_dma(const char* s0, const char* s1, const char* s2, unsigned voff, unsigned d0, unsigned d1,
                                       unsigned d2) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(s0), "s"(s1), "s"(s2), "v"(voff), "s"(d0), "s"(d1), "s"(d2)
      : "memory");
}

// Timing-only knock-outs (tools/ubench/build_alt.sh ... -DUA2_G2_DBG=<bits>; wrong results): 1 no ring refills after the prologue,
// 2 no MFMAs, 4 no fragment reads.
#ifndef UA2_G2_DBG
#define UA2_G2_DBG 0
#endif
// bit 8: cycle stamps of waves 0 (group 0) and 4 (group 1) of workgroup 0 over chunks 8 .. 11 of its loop, read back through
// ua2_g2_stamps (tools/ubench/g2_stamps.py); s_memtime costs ~11 % of a wave's cycles: a build for looking, not for timing
#if UA2_G2_DBG & 8
__device__ unsigned long long g_g2_stamp[2][4][8];
extern "C" int ua2_g2_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g2_stamp), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#define G2_STAMP(slot)                                                                                              \
  do {                                                                                                              \
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wn == 0 && c >= 8 && c < 12)                            \
      g_g2_stamp[grp][c - 8][(slot)] = __builtin_readcyclecounter();                                              \
  } while (0)
#else
#define G2_STAMP(slot) do { } while (0)
#endif

// VAR (same bits, a cost choice): 0 = plain, 1 = s_setprio 1 around the MFMAs (measured 1-3 % behind).  A third form — the requests
// of chunk c + NB - 1 issued inside C(c), spread among its MFMAs — was measured 10 % behind and removed (profiles/r5_notes.md §1).
// workgroup id -> (row-block pm, column-block pn), as ua2_gemm.hip: an XCD gets a contiguous id range, row-blocks fastest inside
// patches of group_m.  `pid` counts over ALL tiles of the problem (a tail launch starts at pid0), `total` = mblocks * nblocks.
__device__ __forceinline__ void g2_tile_of(int pid, const int total, const int mblocks, const int nblocks, const int group_m, int& pm, int& pn) {
  if (total % 8 == 0) pid = (pid & 7) * (total >> 3) + (pid >> 3);
  const int per_group = group_m * nblocks;
  const int group = pid / per_group, first_m = group * group_m;
  const int gsz = min(mblocks - first_m, group_m);
  pm = first_m + (pid % per_group) % gsz;
  pn = (pid % per_group) / gsz;
}

// The part of the RESIDUAL / STORE / SWIGLU / GELU epilogues that runs on 4 consecutive columns n0 .. n0 + 3 of row m: row scale
// (UA2_PRO_SCALED), bias, LayerScale + residual | activation, stores, packed hand-off, scaled-norm hand-over, arg-max partials.
// Called by every lane of a 4-lane group (16-column tile) together: the hand-over and the arg-max exchange across the group; `live`
// = this lane's columns exist (n0 < N) and the row exists.  One definition for the kernel's epilogue and the K-split combines.
template <int EPI>
__device__ __forceinline__ void g2_finish(const ua2_linear_args& a, const int m, const int n0, float4 v0, float4 v1, const float4& cb0, const float4& cb1,
                                          const float4& cos4, const float rs, const float4& res, const bool live, const int forbid) {
  constexpr int KC = 32;
  if (a.prologue == UA2_PRO_SCALED) {
    v0.x = __fmul_rn(v0.x, rs); v0.y = __fmul_rn(v0.y, rs); v0.z = __fmul_rn(v0.z, rs); v0.w = __fmul_rn(v0.w, rs);
    if constexpr (EPI == UA2_EPI_SWIGLU) { v1.x = __fmul_rn(v1.x, rs); v1.y = __fmul_rn(v1.y, rs); v1.z = __fmul_rn(v1.z, rs); v1.w = __fmul_rn(v1.w, rs); }
  }
  if (a.bias) {
    v0.x = __fadd_rn(v0.x, cb0.x); v0.y = __fadd_rn(v0.y, cb0.y); v0.z = __fadd_rn(v0.z, cb0.z); v0.w = __fadd_rn(v0.w, cb0.w);
    if constexpr (EPI == UA2_EPI_SWIGLU) { v1.x = __fadd_rn(v1.x, cb1.x); v1.y = __fadd_rn(v1.y, cb1.y); v1.z = __fadd_rn(v1.z, cb1.z); v1.w = __fadd_rn(v1.w, cb1.w); }
  }
  float4 out = v0;
  if constexpr (EPI == UA2_EPI_RESIDUAL) {
    if (a.out_scale) { out.x = __fmul_rn(cos4.x, v0.x); out.y = __fmul_rn(cos4.y, v0.y); out.z = __fmul_rn(cos4.z, v0.z); out.w = __fmul_rn(cos4.w, v0.w); }
    out.x = __fadd_rn(out.x, res.x); out.y = __fadd_rn(out.y, res.y); out.z = __fadd_rn(out.z, res.z); out.w = __fadd_rn(out.w, res.w);
  } else if constexpr (EPI == UA2_EPI_SWIGLU) {
    out.x = ua2_act_glu(a, v0.x, v1.x); out.y = ua2_act_glu(a, v0.y, v1.y); out.z = ua2_act_glu(a, v0.z, v1.z); out.w = ua2_act_glu(a, v0.w, v1.w);
  } else if constexpr (EPI == UA2_EPI_GELU) {
    out.x = ua2_act_gelu(a, v0.x); out.y = ua2_act_gelu(a, v0.y); out.z = ua2_act_gelu(a, v0.z); out.w = ua2_act_gelu(a, v0.w);
  }
  if (live && a.y) *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n0) = out;
  if constexpr (EPI == UA2_EPI_SWIGLU || EPI == UA2_EPI_GELU) {
    if (live && a.y_packed) store_packed4<UA2_BF16>(a.y_packed, m, n0, a.N / KC, out);
  }
  if constexpr (EPI == UA2_EPI_STORE || EPI == UA2_EPI_RESIDUAL) {
    if (a.y_norm_w) {     // producer half of the scaled-norm hand-over: the 16-column tile = this lane's 4 columns + 3 neighbours
      float s = live ? __fadd_rn(__fmaf_rn(out.x, out.x, __fmul_rn(out.y, out.y)), __fmaf_rn(out.z, out.z, __fmul_rn(out.w, out.w))) : 0.f;
      s = __fadd_rn(s, __shfl_xor(s, 1));
      s = __fadd_rn(s, __shfl_xor(s, 2));
      if (live) {
        if (a.y_ssq && ((n0 >> 2) & 3) == 0) a.y_ssq[(size_t)m * ((a.N + 15) >> 4) + (n0 >> 4)] = s;
        const float4 nw4 = *reinterpret_cast<const float4*>(a.y_norm_w + n0);
        const float4 hh = make_float4(__fmul_rn(out.x, nw4.x), __fmul_rn(out.y, nw4.y), __fmul_rn(out.z, nw4.z), __fmul_rn(out.w, nw4.w));
        if (a.y_h) store_row4<UA2_BF16>(a.y_h, (size_t)m * a.ldh + n0, hh);
        if (a.y_packed) store_packed4<UA2_BF16>(a.y_packed, m, n0, a.N / KC, hh);
      }
    }
  }
  if constexpr (EPI == UA2_EPI_STORE) {
    if (a.part_max) {     // per-16-column (max, index) for the greedy tail, ties -> lowest index, columns below forbid[m] masked (linear_epilogue's rule)
      float bv = -INFINITY;
      int bi = n0;
      if (live) {
        const float vv[4] = {out.x, out.y, out.z, out.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = (n0 + e >= forbid) ? vv[e] : -INFINITY;
          if (t > bv) { bv = t; bi = n0 + e; }              // ascending scan: a tie keeps the lower index
        }
      }
#pragma unroll
      for (int o = 1; o <= 2; o <<= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (live && ((n0 >> 2) & 3) == 0) {
        const int nb = (a.N + 15) / 16;
        a.part_max[(size_t)m * nb + (n0 >> 4)] = bv;
        a.part_idx[(size_t)m * nb + (n0 >> 4)] = bi;
      }
    }
  }
}

// flags: 2 = K slabs over the whole grid (gridDim.y slabs, raw sums into split_ws [S][M][N]); 4 = TAIL launch: workgroup x is tile
// pid0 + x, gridDim.y slabs, raw sums into the compact scratch [S][tail tiles][BMT * 16 rows][256] (gemm2_tail_combine_kernel)
template <int EPI, int BMT, int NB, int VAR>
__global__ __launch_bounds__(512, (BMT == 8 && NB <= 3) ? 4 : 2) void gemm2_kernel(const ua2_linear_args a, const char* __restrict__ apack, const int mblocks,
                                                       const int nblocks, const int group_m, const int flags, const int pid0,
                                                       const float* __restrict__ rstd) {
  constexpr int KC = 32;
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr int WM = BMT / 2;            // row tiles per wave (two groups of four waves split the rows)
  constexpr int WNT = 4 / NT;            // column tiles per wave, per matrix
  constexpr int BNM = 16 / NT;           // column tiles per workgroup, per matrix
  constexpr int TILES = BMT + 16;        // fragment blocks per chunk
  constexpr int LOADS = TILES / 8;       // LDS-DMA requests per wave and chunk
  constexpr int NF = WM + 4;             // fragments a wave reads per chunk
  constexpr bool PRIO = VAR == 1;
  static_assert(TILES % 8 == 0 && (LOADS == 3 || LOADS == 4), "request lists below");
  static_assert((NB - 1) * LOADS <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char g2_smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int nchunks_all = (a.K + KC - 1) / KC;
  const int c_lo = (int)(((long)blockIdx.y * nchunks_all) / gridDim.y), c_hi = (int)(((long)(blockIdx.y + 1) * nchunks_all) / gridDim.y);
  const int nchunks = c_hi - c_lo;
  const int mtiles = (a.M + 15) / 16, ntiles = (a.N + 15) / 16;

  int pm, pn;
  g2_tile_of((int)blockIdx.x + pid0, mblocks * nblocks, mblocks, nblocks, group_m, pm, pn);

  // this wave's LOADS fragment streams: block j * 8 + wave of the chunk
  const char* tb[LOADS];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) {
    const int tile = j * 8 + wave;
    if (tile < BMT) {
      const int mt = min(pm * BMT + tile, mtiles - 1);
      tb[j] = apack + ((size_t)mt * nchunks_all + c_lo) * 1024;
    } else {
      const int idx = tile - BMT, mat = idx / BNM;
      const int nt = min(pn * BNM + idx % BNM, ntiles - 1);
      tb[j] = reinterpret_cast<const char*>(mat ? a.w1 : a.w0) + ((size_t)nt * nchunks_all + c_lo) * 1024;
    }
  }
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned lds0 = g2_lds_addr(g2_smem) + (unsigned)wave * 1024u;
  auto dma = [&](int c_req, int slot) {
    const size_t co = (size_t)min(c_req, nchunks - 1) * 1024;        // past the end: the last chunk again, into a slot nobody reads
    const unsigned d = lds0 + (unsigned)(slot * TILES) * 1024u;
    if constexpr (LOADS == 4) g2_dma(tb[0] + co, tb[1] + co, tb[2] + co, tb[3] + co, voff, d, d + 8192u, d + 16384u, d + 24576u);
    else g2_dma(tb[0] + co, tb[1] + co, tb[2] + co, voff, d, d + 8192u, d + 16384u);
  };

  const u32x4* lfa = reinterpret_cast<const u32x4*>(g2_smem) + (size_t)(grp * WM) * 64 + lane;          // this wave's A blocks of slot 0
  const u32x4* lfb = reinterpret_cast<const u32x4*>(g2_smem) + (size_t)(BMT + wn * WNT) * 64 + lane;    // ... and B blocks (matrix 0)

  f32x4 acc[WM][4];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) acc[mi][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fr[NF];

  // prologue: chunks 0 .. NB - 2 requested, chunk 0 landed
#pragma unroll
  for (int t = 0; t < NB - 1; ++t) dma(t, t);
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NB - 2) * LOADS) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (grp == 1) {                                    // group 1 runs one phase behind
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  // One chunk step of a wave = L(c) then C(c), slot index U = c % NB a compile-time constant (fragment addresses are immediates).
  // Both groups run the SAME instruction stream — the wait for chunk c + 1 sits behind the requests of L(c) in both (group 0 could
  // wait a phase later; a group-dependent wait costs two branches and their VALU -> SALU round trips per chunk: ~200 of the
  // ~1550 cycles a chunk took, cycle stamps in profiles/r5_notes.md §2) — and nothing but the two barriers separates the phases.
  auto step = [&](auto u_c, const int c) {
    constexpr int u = decltype(u_c)::value;
    // ---- L(c) ----
    G2_STAMP(0);
    if constexpr (!(UA2_G2_DBG & 4)) {
#pragma unroll
      for (int mi = 0; mi < WM; ++mi) fr[mi] = lfa[(size_t)(u * TILES + mi) * 64];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ni = 0; ni < WNT; ++ni) fr[WM + t * WNT + ni] = lfb[(size_t)(u * TILES + t * BNM + ni) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    G2_STAMP(1);
    if constexpr (!(UA2_G2_DBG & 1)) dma(c + NB - 1, (u + NB - 1) % NB);
    G2_STAMP(2);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((UA2_G2_DBG & 1) ? 0 : (NB - 2) * LOADS) : "memory");
    G2_STAMP(3);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    G2_STAMP(4);
    // ---- C(c) ----
    if constexpr (!(UA2_G2_DBG & 2)) {
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
          acc[mi][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[mi]), __builtin_bit_cast(bf16x8, fr[WM + ci]),
                                                                acc[mi][ci], 0, 0, 0);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    G2_STAMP(5);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    G2_STAMP(7);
  };
  using std::integral_constant;
  int c0 = 0;
  for (; c0 + NB <= nchunks; c0 += NB) {             // whole turns of the ring: no per-step bound checks
    step(integral_constant<int, 0>{}, c0);
    step(integral_constant<int, 1>{}, c0 + 1);
    if constexpr (NB > 2) step(integral_constant<int, 2>{}, c0 + 2);
    if constexpr (NB > 3) step(integral_constant<int, 3>{}, c0 + 3);
    if constexpr (NB > 4) step(integral_constant<int, 4>{}, c0 + 4);
    if constexpr (NB > 5) step(integral_constant<int, 5>{}, c0 + 5);
  }
  {                                                  // the last, partial turn
    const int r = nchunks - c0;
    if (r > 0) step(integral_constant<int, 0>{}, c0);
    if (r > 1) step(integral_constant<int, 1>{}, c0 + 1);
    if constexpr (NB > 3) { if (r > 2) step(integral_constant<int, 2>{}, c0 + 2); }
    if constexpr (NB > 4) { if (r > 3) step(integral_constant<int, 3>{}, c0 + 3); }
    if constexpr (NB > 5) { if (r > 4) step(integral_constant<int, 4>{}, c0 + 4); }
  }
  if (grp == 0) {                                    // group 1's last C phase
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail requests: nothing may land once the epilogue reuses the ring
  __syncthreads();

  // ---- epilogues: the wave's patch goes through its share of the ring PM row tiles at a time ----
  constexpr int SHARE = NB * TILES * 1024 / 8;                       // bytes of LDS per wave
  constexpr int PM0 = (SHARE / 4096 >= WM) ? WM : ((SHARE / 4096 >= WM / 2) ? WM / 2 : WM / 4);
  // RESIDUAL requests every residual piece of a pass before its first store (PM x 4 float4 per lane): two row tiles per pass keep the
  // 256-row-tile form inside its registers now that the epilogue also carries the hand-over
  // (one per pass in the 128-register form, whose unrolled pass would not fit otherwise)
  constexpr bool kSmallRegs = BMT == 8 && NB <= 3 && (EPI == UA2_EPI_RESIDUAL || EPI == UA2_EPI_STORE);   // the hand-over / arg-max epilogues in 128 registers
  // (the 128-row form with the deep ring — one workgroup per CU, the small-M launches of the DiT — has the registers for a whole-patch
  // pass: every pass more is one more exposed residual round trip on a launch whose fixed cost is half its time, profiles/r6_notes.md §9)
  constexpr int PM = kSmallRegs ? 1 : ((EPI == UA2_EPI_RESIDUAL && BMT == 16 && PM0 > 2) ? 2 : PM0);
  static_assert(PM >= 1 && WM % PM == 0 && PM * 4096 <= SHARE, "patch does not fit the wave's share of the ring");
  float* patch = reinterpret_cast<float*>(g2_smem + (size_t)wave * SHARE);
  const int colq = lane & 15, gq = lane >> 4;
  const int mwave = (pm * BMT + grp * WM) * 16;                      // first row of this wave's patch

  auto park = [&](int p0, auto colmap) {                             // rows of tiles p0 .. p0 + PM - 1 into the patch
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      if (mi < p0 || mi >= p0 + PM) continue;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[((mi - p0) * 16 + 4 * gq + r) * 64 + colmap(ci)] = acc[mi][ci][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto done = [&]() {                                                // the patch is re-used by the next pass: reads before writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  if constexpr (EPI == UA2_EPI_QKV_ROPE) {
    if (a.rope_mode == UA2_ROPE_HALF_SPLIT) {
      // half-split rotation, no bias (the LM): head size 128 (trunk: the workgroup's 256 columns are two heads, a wave holds half a
      // head) or 64 (depth decoder: four heads, a wave holds one).  In the packed (permuted) order tile r of a head carries dims
      // [8r, 8r + 8) and their rotation partners [hs/2 + 8r, ..): a wave's four tiles are 32 low dims and their 32 partners.  Parked in
      // natural dim order ([0, 32): the low dims, [32, 64): the partners) a lane owns 4 consecutive dims of a row and finds its
      // rotation partner 32 columns away.  Same operations as ua2_gemm.hip's staged form.
      const int hs = a.kv.head_size, half = hs >> 1;            // 128 / 64 (the launcher admits nothing else)
      const int h = (hs == 128) ? pn * 2 + (wn >> 1) : pn * 4 + wn, half_id = (hs == 128) ? (wn & 1) : 0;
      const bool is_q = h < a.kv.n_head, is_k = !is_q && h < a.kv.n_head + a.kv.n_kv;
      const bool rot = is_q || is_k;
      const int kvh = is_q ? 0 : (is_k ? h - a.kv.n_head : h - a.kv.n_head - a.kv.n_kv);
      const int j = lane & 15, jj = j & 7;
      const bool hi = j >= 8;
      const int d0 = 32 * half_id + 4 * jj;
      const bool live = h < a.kv.n_head + 2 * a.kv.n_kv;
#pragma unroll 1
      for (int p0 = 0; p0 < WM; p0 += PM) {
        park(p0, [&](int ci) { return (colq < 8) ? ci * 8 + colq : 32 + ci * 8 + (colq - 8); });
        if (live) {
#pragma unroll 1
          for (int it = 0; it < PM * 4; ++it) {
            const int prow = it * 4 + gq;
            const int m = mwave + p0 * 16 + prow;
            if (m >= a.M) continue;
            float4 own = *reinterpret_cast<const float4*>(patch + prow * 64 + (hi ? 32 : 0) + 4 * jj);
            float4 oth = *reinterpret_cast<const float4*>(patch + prow * 64 + (hi ? 0 : 32) + 4 * jj);
            if (a.prologue == UA2_PRO_SCALED) {          // the row scale first, as linear_epilogue does
              const float rs = rstd[m];
              own.x = __fmul_rn(own.x, rs); own.y = __fmul_rn(own.y, rs); own.z = __fmul_rn(own.z, rs); own.w = __fmul_rn(own.w, rs);
              oth.x = __fmul_rn(oth.x, rs); oth.y = __fmul_rn(oth.y, rs); oth.z = __fmul_rn(oth.z, rs); oth.w = __fmul_rn(oth.w, rs);
            }
            const int pos = a.row_pos[m];
            float4 out = own;
            if (rot) {
              const float4 cs = *reinterpret_cast<const float4*>(a.rope_cos + (size_t)pos * half + d0);
              const float4 sn = *reinterpret_cast<const float4*>(a.rope_sin + (size_t)pos * half + d0);
              if (!hi) {
                out.x = __fadd_rn(__fmul_rn(own.x, cs.x), __fmul_rn(-oth.x, sn.x)); out.y = __fadd_rn(__fmul_rn(own.y, cs.y), __fmul_rn(-oth.y, sn.y));
                out.z = __fadd_rn(__fmul_rn(own.z, cs.z), __fmul_rn(-oth.z, sn.z)); out.w = __fadd_rn(__fmul_rn(own.w, cs.w), __fmul_rn(-oth.w, sn.w));
              } else {
                out.x = __fadd_rn(__fmul_rn(own.x, cs.x), __fmul_rn(oth.x, sn.x)); out.y = __fadd_rn(__fmul_rn(own.y, cs.y), __fmul_rn(oth.y, sn.y));
                out.z = __fadd_rn(__fmul_rn(own.z, cs.z), __fmul_rn(oth.z, sn.z)); out.w = __fadd_rn(__fmul_rn(own.w, cs.w), __fmul_rn(oth.w, sn.w));
              }
            }
            const int dd = (hi ? half : 0) + d0;
            if (is_q) {
              *reinterpret_cast<float4*>(a.q_out + (size_t)m * a.kv.n_head * hs + (size_t)h * hs + dd) = out;
            } else {
              const int page = a.kv.page_table[(size_t)kv_table_row(a, m) * a.kv.max_pages + ua2_page_slot(a.kv, pos)];
              const size_t base = (((size_t)page * a.kv.n_kv + kvh) * UA2_PAGE + (pos % UA2_PAGE)) * hs + dd;
              uint2 pk;
              pk.x = (unsigned)f2bf(out.x) | ((unsigned)f2bf(out.y) << 16);
              pk.y = (unsigned)f2bf(out.z) | ((unsigned)f2bf(out.w) << 16);
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(is_k ? a.kv.k_pool : a.kv.v_pool) + base) = pk;
            }
          }
        }
        done();
      }
      return;
    } else {
      // no rotation, head size 64, optional bias (the DiT's fused q|k|v): a wave's 64 columns are one head
      const int hs = 64;
      const int h = pn * 4 + wn;
      const bool live = h * hs < a.N;
      const bool is_q = h < a.kv.n_head, is_k = !is_q && h < a.kv.n_head + a.kv.n_kv;
      const int kvh = is_q ? 0 : (is_k ? h - a.kv.n_head : h - a.kv.n_head - a.kv.n_kv);
      const int d0 = 4 * colq;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias && live) b4 = *reinterpret_cast<const float4*>(a.bias + (size_t)h * hs + d0);
#pragma unroll 1
      for (int p0 = 0; p0 < WM; p0 += PM) {
        park(p0, [&](int ci) { return ci * 16 + colq; });
        if (live) {
#pragma unroll 1
          for (int it = 0; it < PM * 4; ++it) {
            const int prow = it * 4 + gq;
            const int m = mwave + p0 * 16 + prow;
            if (m >= a.M) continue;
            float4 out = *reinterpret_cast<const float4*>(patch + prow * 64 + d0);
            if (a.bias) { out.x = __fadd_rn(out.x, b4.x); out.y = __fadd_rn(out.y, b4.y); out.z = __fadd_rn(out.z, b4.z); out.w = __fadd_rn(out.w, b4.w); }
            if (is_q) {
              *reinterpret_cast<float4*>(a.q_out + (size_t)m * a.kv.n_head * hs + (size_t)h * hs + d0) = out;
            } else {
              const int pos = a.row_pos[m];
              const int page = a.kv.page_table[(size_t)kv_table_row(a, m) * a.kv.max_pages + ua2_page_slot(a.kv, pos)];
              const size_t base = (((size_t)page * a.kv.n_kv + kvh) * UA2_PAGE + (pos % UA2_PAGE)) * hs + d0;
              uint2 pk;
              pk.x = (unsigned)f2bf(out.x) | ((unsigned)f2bf(out.y) << 16);
              pk.y = (unsigned)f2bf(out.z) | ((unsigned)f2bf(out.w) << 16);
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(is_k ? a.kv.k_pool : a.kv.v_pool) + base) = pk;
            }
          }
        }
        done();
      }
      return;
    }
  } else {
    // STORE / RESIDUAL / SWIGLU / GELU
    constexpr int SPAN = WNT * 16;                       // output columns of a wave: 64 (32 for SWIGLU)
    constexpr int LPR = SPAN / 4;                        // lanes per row
    constexpr int RPI = 64 / LPR;                        // rows per iteration
    constexpr int ITERS = PM * 16 / RPI;
    const bool slab = (flags & 2) != 0, tail = (flags & 4) != 0;   // K split: raw partial sums into split_ws
    const int j = lane & (LPR - 1), rsub = lane / LPR;
    const int n0 = (pn * BNM + wn * WNT) * 16 + 4 * j;   // this lane's 4 columns (of each matrix)
    const bool clive = n0 < a.N;                         // (N % 4 == 0: a lane's four columns exist together)
    if (n0 - 4 * j >= a.N) return;                       // the wave's whole span is past N (wave-uniform)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr bool PRE = !(BMT == 8 && NB <= 3);         // the two-workgroups-per-CU form has 128 registers: request per row instead
    float4 cb0 = zero4, cb1 = zero4, cos4 = make_float4(1.f, 1.f, 1.f, 1.f);      // per-lane column constants
    if (clive) {
      if (a.bias) cb0 = *reinterpret_cast<const float4*>(a.bias + n0);
      if constexpr (NT == 2) { if (a.bias && a.bias1) cb1 = *reinterpret_cast<const float4*>(a.bias1 + n0); }
      if constexpr (EPI == UA2_EPI_RESIDUAL) { if (a.out_scale) cos4 = *reinterpret_cast<const float4*>(a.out_scale + n0); }
    }
    auto pass = [&](auto p0v) {                          // one pass of PM row tiles; p0v: int, or an integral_constant (fully static accumulator indices)
      const int p0 = p0v;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);   // (a local: capturing the outer constant by reference gave it a stack slot)
      park(p0, [&](int ci) { return ci * 16 + colq; });
      {
        const int mbase = mwave + p0 * 16;
        float4 res[PRE ? ITERS : 1];
        if constexpr (EPI == UA2_EPI_RESIDUAL && PRE) {  // every residual piece of the pass requested before the first store
          if (!slab && !tail && clive) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
              const int m = mbase + it * RPI + rsub;
              res[it] = *reinterpret_cast<const float4*>(a.resid + (size_t)min(m, a.M - 1) * a.ldr + n0);
            }
          }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int prow = it * RPI + rsub;
          const int m = mbase + prow;
          if (m >= a.M) continue;                        // uniform over the row's lanes
          float4 v0 = *reinterpret_cast<const float4*>(patch + prow * 64 + 4 * j);
          if constexpr (EPI == UA2_EPI_RESIDUAL) {
            if (slab) {
              if (clive) *reinterpret_cast<float4*>(a.split_ws + ((size_t)blockIdx.y * a.M + m) * a.N + n0) = v0;
              continue;
            }
            if (tail) {                                  // compact scratch: [slab][tail tile][row of the tile][256 columns]
              const int trow = (grp * WM + p0) * 16 + prow;
              *reinterpret_cast<float4*>(a.split_ws + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BMT * 16) + trow) * 256 + wn * 64 + 4 * j) = v0;
              continue;
            }
          }
          float4 v1 = z4;
          if constexpr (NT == 2) v1 = *reinterpret_cast<const float4*>(patch + prow * 64 + SPAN + 4 * j);
          float4 r4 = z4;
          if constexpr (EPI == UA2_EPI_RESIDUAL) {
            if constexpr (PRE) r4 = res[it];
            else if (clive) r4 = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n0);
          }
          const float rs = (a.prologue == UA2_PRO_SCALED) ? rstd[m] : 1.f;
          int forbid = 0;
          if constexpr (EPI == UA2_EPI_STORE) { if (a.part_max && a.forbid) forbid = a.forbid[m]; }
          g2_finish<EPI>(a, m, n0, v0, v1, cb0, cb1, cos4, rs, r4, clive, forbid);
        }
      }
      done();
    };
    if constexpr (kSmallRegs) {
      // the 128-register form walks its four row tiles one per pass; written out, so that no accumulator is selected at run time
      // (as a loop the compiler parked the accumulators in scratch: 208 bytes per lane)
      static_assert(WM == 4 && PM == 1, "passes below");
      using std::integral_constant;
      pass(integral_constant<int, 0>{}); pass(integral_constant<int, 1>{}); pass(integral_constant<int, 2>{}); pass(integral_constant<int, 3>{});
    } else {
#pragma unroll 1
      for (int p0 = 0; p0 < WM; p0 += PM) pass(p0);
    }
  }
}

// y = resid + out_scale (.) ((((s0 + s1) + s2) + s3) + bias): the slabs of a K split in index order (as ua2_gemm.hip's combine),
// then everything the RESIDUAL epilogue does (g2_finish: stores, scaled-norm hand-over).  N % 64 == 0: whole 4-lane groups.
__global__ __launch_bounds__(256) void gemm2_combine_kernel(const ua2_linear_args a, const int slabs) {
  const int n4 = a.N >> 2;
  const size_t total = (size_t)a.M * n4, slab_elems = (size_t)a.M * a.N;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < total; i0 += (size_t)gridDim.x * blockDim.x) {
    const size_t i = i0 + threadIdx.x;
    const bool live = i < total;                           // (total % 4 == 0 and blockDim % 4 == 0: a 4-lane group is live together)
    const size_t ic = live ? i : total - 1;
    const int m = (int)(ic / n4), n = (int)(ic - (size_t)m * n4) * 4;
    const float* p = a.split_ws + (size_t)m * a.N + n;
    float4 t = *reinterpret_cast<const float4*>(p);
    for (int k = 1; k < slabs; ++k) {
      const float4 u = *reinterpret_cast<const float4*>(p + (size_t)k * slab_elems);
      t.x = __fadd_rn(t.x, u.x); t.y = __fadd_rn(t.y, u.y); t.z = __fadd_rn(t.z, u.z); t.w = __fadd_rn(t.w, u.w);
    }
    float4 cb0 = zero4, cos4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.bias) cb0 = *reinterpret_cast<const float4*>(a.bias + n);
    if (a.out_scale) cos4 = *reinterpret_cast<const float4*>(a.out_scale + n);
    const float4 r = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n);
    g2_finish<UA2_EPI_RESIDUAL>(a, m, n, t, zero4, cb0, zero4, cos4, 1.f, r, live, 0);
  }
}

// The tail launch's slabs (compact scratch [S][ntail][rows][256]) -> the tiles pid0 .. pid0 + ntail - 1: a workgroup = 4 rows of a tile
__global__ __launch_bounds__(256) void gemm2_tail_combine_kernel(const ua2_linear_args a, const int slabs, const int ntail, const int pid0, const int mblocks,
                                                                 const int nblocks, const int group_m, const int bmt) {
  const int rows = bmt * 16;
  const int t = blockIdx.x / (rows / 4), r = (blockIdx.x % (rows / 4)) * 4 + (threadIdx.x >> 6), c4 = threadIdx.x & 63;
  int pm, pn;
  g2_tile_of(pid0 + t, mblocks * nblocks, mblocks, nblocks, group_m, pm, pn);
  const int m = pm * rows + r, n = pn * 256 + c4 * 4;
  const bool rlive = m < a.M;                              // uniform over the row's 64 threads
  if (!rlive) return;
  const bool live = n < a.N;
  const float* p = a.split_ws + (((size_t)t) * rows + r) * 256 + c4 * 4;
  const size_t slab_elems = (size_t)ntail * rows * 256;
  float4 s4 = *reinterpret_cast<const float4*>(p);
  for (int k = 1; k < slabs; ++k) {
    const float4 u = *reinterpret_cast<const float4*>(p + (size_t)k * slab_elems);
    s4.x = __fadd_rn(s4.x, u.x); s4.y = __fadd_rn(s4.y, u.y); s4.z = __fadd_rn(s4.z, u.z); s4.w = __fadd_rn(s4.w, u.w);
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 cb0 = zero4, cos4 = make_float4(1.f, 1.f, 1.f, 1.f), r4 = zero4;
  if (live) {
    if (a.bias) cb0 = *reinterpret_cast<const float4*>(a.bias + n);
    if (a.out_scale) cos4 = *reinterpret_cast<const float4*>(a.out_scale + n);
    r4 = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n);
  }
  g2_finish<UA2_EPI_RESIDUAL>(a, m, n, s4, zero4, cb0, zero4, cos4, 1.f, r4, live, 0);
}

// LayerNorm hand-over (ua2hip.h y_ln_w).  slabs > 0: the K-split combine and the next GEMM's LayerNorm prep in ONE pass — y = resid +
// out_scale (.) (sum of the slabs in index order + bias), stored, and RNE_bf16(LayerNorm(y) * w + b) written in fragment order (the
// DiT's o-projection / FF2: the prep launch of FF1 / the next block's q|k|v disappears).  slabs == 0: y is already complete (no K split,
// or a tail split): the LayerNorm pass alone.  A WORKGROUP per row — one 4-column piece per thread, N / 4 threads rounded up to whole
// waves, statistics through LDS; norm_stat / norm_apply's formulas.  (A wave per row, six pieces per lane, was measured first: 14 us
// for 1000 rows against ~5.5 — each of a row's steps is a memory round trip (the slabs were written by other XCDs' workgroups: every
// load misses this XCD's L2) and with one wave per SIMD nothing overlapped them: DiT step 5.17 vs 4.89 ms, profiles/r6_notes.md §6.)
__global__ __launch_bounds__(512) void gemm2_combine_ln_row_kernel(const ua2_linear_args a, const int slabs) {
  __shared__ float red_s[2][8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int n = 4 * tid;
  const bool live = n < a.N;
  const int nc = min(n, a.N - 4);
  const size_t slab_elems = (size_t)a.M * a.N;
  float4 t;
  if (slabs > 0) {
    const float* q = a.split_ws + (size_t)m * a.N + nc;
    t = *reinterpret_cast<const float4*>(q);
    float4 u[3];
#pragma unroll
    for (int k = 1; k < 4; ++k) u[k - 1] = (k < slabs) ? *reinterpret_cast<const float4*>(q + (size_t)k * slab_elems) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 r = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + nc);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      if (k >= slabs) break;
      t.x = __fadd_rn(t.x, u[k - 1].x); t.y = __fadd_rn(t.y, u[k - 1].y); t.z = __fadd_rn(t.z, u[k - 1].z); t.w = __fadd_rn(t.w, u[k - 1].w);
    }
    if (a.bias) {
      const float4 b = *reinterpret_cast<const float4*>(a.bias + nc);
      t.x = __fadd_rn(t.x, b.x); t.y = __fadd_rn(t.y, b.y); t.z = __fadd_rn(t.z, b.z); t.w = __fadd_rn(t.w, b.w);
    }
    if (a.out_scale) {
      const float4 g = *reinterpret_cast<const float4*>(a.out_scale + nc);
      t.x = __fmul_rn(g.x, t.x); t.y = __fmul_rn(g.y, t.y); t.z = __fmul_rn(g.z, t.z); t.w = __fmul_rn(g.w, t.w);
    }
    t.x = __fadd_rn(t.x, r.x); t.y = __fadd_rn(t.y, r.y); t.z = __fadd_rn(t.z, r.z); t.w = __fadd_rn(t.w, r.w);
    if (live) *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = t;
  } else {
    t = *reinterpret_cast<const float4*>(a.y + (size_t)m * a.ldy + nc);
  }
  if (!live) t = make_float4(0.f, 0.f, 0.f, 0.f);
  // the norm weights of this thread's columns travel while the statistics are reduced
  const float4 w = *reinterpret_cast<const float4*>(a.y_ln_w + nc);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.y_ln_b) b = *reinterpret_cast<const float4*>(a.y_ln_b + nc);
  float sm = sum4(0.f, t), ss = sumsq4(0.f, t);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { sm += __shfl_xor(sm, o); ss += __shfl_xor(ss, o); }
  if (lane == 0) { red_s[0][wave] = sm; red_s[1][wave] = ss; }
  __syncthreads();
  sm = 0.f; ss = 0.f;
  for (int i = 0; i < nw; ++i) { sm += red_s[0][i]; ss += red_s[1][i]; }       // every thread, wave order
  const float mean = sm / (float)a.N;
  const float rstd = 1.0f / sqrtf(fmaxf(__fsub_rn(ss / (float)a.N, __fmul_rn(mean, mean)), 0.f) + a.y_ln_eps);
  if (!live) return;
  float4 o4;
  o4.x = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(t.x, mean), rstd), w.x), b.x);
  o4.y = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(t.y, mean), rstd), w.y), b.y);
  o4.z = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(t.z, mean), rstd), w.z), b.z);
  o4.w = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(t.w, mean), rstd), w.w), b.w);
  store_packed4<UA2_BF16>(a.y_packed, m, n, a.N / 32, o4);
}

// UA2_PRO_SCALED consumers: one scale per row, reduced from the producer's per-16-column partials in the one order every kernel uses
// (scaled_rstd_row: 16 interleaved chains + butterfly) — a launch of M / 16 workgroups in front of the GEMM instead of 16 lanes per
// row and column block inside it
__global__ __launch_bounds__(256) void gemm2_rstd_kernel(const ua2_linear_args a, float* __restrict__ rstd) {
  const int m = blockIdx.x * 16 + (threadIdx.x >> 4);
  const float rs = scaled_rstd_row(a, min(m, a.M - 1), threadIdx.x & 15);
  if (m < a.M && (threadIdx.x & 15) == 0) rstd[m] = rs;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tuning / A-B knobs: read once (ua2_common.h Ua2EnvInt; ua2_debug_refresh_env re-reads them — the tests toggle some)
Ua2EnvInt g_group_m{"UA2_GEMM2_GROUP_M", 8}, g_ks_min_chunks{"UA2_GEMM2_KSPLIT_MIN_CHUNKS", 96}, g_ks_max_grid{"UA2_GEMM2_KSPLIT_MAX_GRID", 128},
    g_no_ksplit{"UA2_GEMM_NO_KSPLIT", 0}, g_bmt{"UA2_GEMM2_BMT", 0}, g_deep_max_grid{"UA2_GEMM2_DEEP_MAX_GRID", 256}, g_no_deep{"UA2_GEMM2_NO_DEEP", 0},
    g_off{"UA2_GEMM2_OFF", 0}, g_min_rows{"UA2_GEMM2_MIN_ROWS", 256}, g_no_tail{"UA2_GEMM2_NO_TAIL", 0}, g_ks_max{"UA2_GEMM2_KSPLIT_MAX", 4}, g_tail_min_chunks{"UA2_GEMM2_TAIL_MIN_CHUNKS", 128}, g_no_new{"UA2_GEMM2_R5_FORMS", 0};

// Instantiations: BMT = 16 / four slots (one workgroup per CU), BMT = 8 / three slots (two per CU, 128 registers) and BMT = 8 / six
// slots (small grids), all without s_setprio.  -DUA2_G2_EXPERIMENTS adds the s_setprio variant and free choice of the ring behind
// the UA2_GEMM2_VAR / UA2_GEMM2_NB hooks (tools/ubench/gemm2_variants.py).
template <int EPI>
int launch2(const ua2_linear_args& a, hipStream_t s, const float* rstd, const bool dry = false) {
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr int BNM = 16 / NT;
  const int mtiles = ua2_ceil_div(a.M, 16), ntiles = ua2_ceil_div(a.N, 16), nblocks = ua2_ceil_div(ntiles, BNM);
  const int nchunks = ua2_ceil_div(a.K, 32);
  const int group_m = std::max(1, g_group_m.get());
  // Tile choice (a cost model fitted on tools/ubench/gemm2_variants.py, profiles/r5_gemm2_variants.txt): 256-row tiles when they
  // keep >= 70 % of the CU slots of their last round busy (and fill half the device at all); else 128-row tiles, two workgroups per
  // CU, when there are >= 128 of them (K slabs included); else the launch is too small for either and goes back to ua2_gemm.hip
  // (M = 1000 x N = 1536: 48 tiles).
  const int64_t t16 = (int64_t)ua2_ceil_div(mtiles, 16) * nblocks, t8 = (int64_t)ua2_ceil_div(mtiles, 8) * nblocks;
  auto slabs_for = [&](int64_t grid) {
    if constexpr (EPI == UA2_EPI_RESIDUAL) {
      // (with the LayerNorm hand-over the combine also replaces the consumer's prep launch: the split pays from 36 chunks of K — the DiT's
      // o-projection, K = 1536 — where without it it does not, profiles/r4_notes.md §13)
      const int min_chunks = a.y_ln_w ? std::min(36, g_ks_min_chunks.get()) : g_ks_min_chunks.get();
      if (a.split_ws && nchunks >= min_chunks && grid <= g_ks_max_grid.get() && !g_no_ksplit.set()) {
        const int want = (int)std::min<int64_t>(std::min(g_ks_max.get(), nchunks / 12), (256 + grid - 1) / grid);
        const int fit = (int)std::min<size_t>(4, a.split_ws_bytes / ((size_t)a.M * a.N * sizeof(float)));
        return std::max(1, std::min(want, fit));
      }
    }
    return 1;
  };
  // TAIL SPLIT (RESIDUAL with scratch): 256-row tiles whose last round would fill less than 70 % of the CUs — the trunk's o- and
  // down-projection at 6272 rows are 25 x 12 = 300 tiles: a second round of 44 — run as whole rounds + the remaining tiles cut
  // into K slabs on a second launch (+ a combine over those tiles only): 334 -> ~230 us for the down-projection.
  int tail = 0, tail_ks = 1;
  if constexpr (EPI == UA2_EPI_RESIDUAL) {
    const int64_t rem = t16 % 256;
    // (long K only: at K = 3072 the o-projection's three launches — 130 us — lose to the 128-row tiles' 121 us, profiles/r6_notes.md §3)
    if (!g_bmt.get() && !g_no_tail.get() && !g_no_ksplit.set() && a.split_ws && nchunks >= g_tail_min_chunks.get() && t16 > 256 && rem > 0 && rem * 10 < 256 * 7) {
      int ks = (int)std::min<int64_t>(8, 256 / rem);
      while (ks > 1 && nchunks / ks < 12) --ks;                                  // a slab must keep the ring busy (>= 12 chunks)
      while (ks > 1 && (size_t)ks * rem * 256 * 256 * sizeof(float) > a.split_ws_bytes) --ks;
      if (ks >= 2) { tail = (int)rem; tail_ks = ks; }
    }
  }
  int bmt = g_bmt.get();
  if (!bmt) {
    const double e16 = (double)t16 / (double)(((t16 + 255) / 256) * 256);
    if (tail || (t16 >= 128 && e16 >= 0.70)) bmt = 16;
    else if (t8 * slabs_for(t8) >= 128) bmt = 8;
    else return 1;
  }
  if (dry) return 0;                                   // the launch fits: the caller may issue what has to precede it
  const int mblocks = ua2_ceil_div(mtiles, bmt);
  const int64_t grid_all = (int64_t)mblocks * nblocks;
  const int64_t grid1 = grid_all - tail;
  const int ks = tail ? 1 : slabs_for(grid_all), flags = ks > 1 ? 2 : 0;
  const char* ap = reinterpret_cast<const char*>(a.x_packed ? a.x_packed : a.workspace);
  auto go = [&](auto bmt_c, auto nb_c, auto var_c, int64_t gx, int gy, int fl, int pid0) {
    constexpr int B = decltype(bmt_c)::value, NBUF = decltype(nb_c)::value, V = decltype(var_c)::value;
    constexpr auto kern = gemm2_kernel<EPI, B, NBUF, V>;
    ua2_allow_big_lds<kern>();
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(512), (size_t)NBUF * (B + 16) * 1024, s, a, ap, mblocks, nblocks, group_m, fl, pid0, rstd);
    ua2_count_launch(UA2_CNT_GEMM2);
  };
  using std::integral_constant;
  auto launch_form = [&](int64_t gx, int gy, int fl, int pid0) {
#ifdef UA2_G2_EXPERIMENTS
    static Ua2EnvInt g_var{"UA2_GEMM2_VAR", 0}, g_nb{"UA2_GEMM2_NB", 0};
    const int var = g_var.get() == 1, nbx = g_nb.get();
    auto pick_var = [&](auto bmt_c, auto nb_c) {
      if (var) go(bmt_c, nb_c, integral_constant<int, 1>{}, gx, gy, fl, pid0); else go(bmt_c, nb_c, integral_constant<int, 0>{}, gx, gy, fl, pid0);
    };
    if (bmt == 16) pick_var(integral_constant<int, 16>{}, integral_constant<int, 4>{});
    else if (nbx == 6) pick_var(integral_constant<int, 8>{}, integral_constant<int, 6>{});
    else pick_var(integral_constant<int, 8>{}, integral_constant<int, 3>{});
#else
    // 128-row tiles: three slots and two workgroups per CU when the grid has more workgroups than CUs; six slots (five chunks in flight)
    // when every workgroup has a CU to itself anyway — the small launches of the DiT's single window start on weights that are in no
    // cache, and with two chunks in flight a workgroup advances one chunk per HBM round trip (measured in situ: profiles/r5_notes.md §3)
    const bool deep = gx * gy <= g_deep_max_grid.get() && !g_no_deep.get();
    if (bmt == 16) go(integral_constant<int, 16>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, gx, gy, fl, pid0);
    else if (deep) go(integral_constant<int, 8>{}, integral_constant<int, 6>{}, integral_constant<int, 0>{}, gx, gy, fl, pid0);
    else go(integral_constant<int, 8>{}, integral_constant<int, 3>{}, integral_constant<int, 0>{}, gx, gy, fl, pid0);
#endif
  };
  launch_form(grid1, ks, flags, 0);
  auto ln_pass = [&](int slabs) {                       // ua2hip.h y_ln_w: combine + LayerNorm prep in one launch (slabs > 0), or the LayerNorm pass alone
    hipLaunchKernelGGL(gemm2_combine_ln_row_kernel, dim3((unsigned)a.M), dim3((unsigned)(ua2_ceil_div(a.N / 4, 64) * 64)), 0, s, a, slabs);
  };
  bool ln_done = false;
  if constexpr (EPI == UA2_EPI_RESIDUAL) {
    if (flags & 2) {
      if (a.y_ln_w) { ln_pass(ks); ln_done = true; }
      else {
        const size_t total4 = (size_t)a.M * (a.N / 4);
        hipLaunchKernelGGL(gemm2_combine_kernel, dim3((unsigned)std::min<size_t>((total4 + 255) / 256, 2048)), dim3(256), 0, s, a, ks);
      }
    }
    if (tail) {
      launch_form(tail, tail_ks, 4, (int)grid1);
      hipLaunchKernelGGL(gemm2_tail_combine_kernel, dim3((unsigned)(tail * (bmt * 16 / 4))), dim3(256), 0, s, a, tail_ks, tail, (int)grid1, mblocks, nblocks, group_m, bmt);
    }
    if (a.y_ln_w && !ln_done) ln_pass(0);
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// 0 = launched, 1 = this launch is outside the kernel's forms (the caller goes on to ua2_gemm.hip's kernels).  The operand is
// already packed (x_packed, or the prep launch into workspace).
static int gemm2_try(const ua2_linear_args& a, hipStream_t s, const bool dry_run) {
  if (a.dtype != UA2_BF16 || g_off.set()) return 1;
  if (a.M < g_min_rows.get() || a.K % 32 != 0) return 1;
  const bool glu = a.epilogue == UA2_EPI_SWIGLU;
  const bool r5 = g_no_new.set();                               // A/B hook: only the launches round 5's kernel took
  if (r5 && (a.prologue == UA2_PRO_SCALED || a.part_max || a.y_norm_w || a.N % (glu ? 32 : 64) != 0)) return 1;
  // columns: whole wave spans (64; 32 per matrix for SWIGLU) — except STORE, whose per-lane column guard takes any N % 4 == 0
  // (audio_head: 12 296)
  if (a.epilogue == UA2_EPI_STORE ? (a.N % 4 != 0) : (a.N % (glu ? 32 : 64) != 0)) return 1;
  if (a.bias && !aligned16(a.bias)) return 1;
  if (glu && a.bias1 && !aligned16(a.bias1)) return 1;
  if (a.part_max && a.epilogue != UA2_EPI_STORE) return 1;
  if (a.y_norm_w) {                                             // producer half of the hand-over: RESIDUAL / STORE, whole 16-column tiles
    if ((a.epilogue != UA2_EPI_RESIDUAL && a.epilogue != UA2_EPI_STORE) || a.N % 16 != 0 || !a.y_ssq || !aligned16(a.y_norm_w)) return 1;
    if ((a.y_h && (a.ldh % 4 != 0 || (reinterpret_cast<uintptr_t>(a.y_h) & 7))) || (a.y_packed && !aligned16(a.y_packed))) return 1;
  }
  if (a.y_ln_w) {                                               // LayerNorm hand-over: RESIDUAL, a wave per row of up to 2048 columns
    if (a.epilogue != UA2_EPI_RESIDUAL || a.y_norm_w || a.N > 2048 || a.N % 32 != 0 || !a.y_packed || !aligned16(a.y_packed) || !aligned16(a.y_ln_w) ||
        (a.y_ln_b && !aligned16(a.y_ln_b)))
      return 1;
  }
  const float* rstd = nullptr;
  if (a.prologue == UA2_PRO_SCALED) {                           // consumer half: one scale per row, reduced by a launch in front, into the workspace
    if (!a.x_packed || !a.x_ssq || !a.workspace || a.workspace == a.x_packed || a.workspace_bytes < (size_t)a.M * sizeof(float)) return 1;
    if (a.epilogue == UA2_EPI_QKV_ROPE && a.rope_mode != UA2_ROPE_HALF_SPLIT) return 1;     // (the un-rotated q|k|v form writes un-scaled sums)
    rstd = reinterpret_cast<const float*>(a.workspace);
  }
  auto with_rstd = [&](auto launch) -> int {
    if (dry_run) return launch(true);
    if (!rstd) return launch(false);
    if (launch(true)) return 1;                                 // the grid rule turns the problem down: nothing issued
    hipLaunchKernelGGL(gemm2_rstd_kernel, dim3(ua2_ceil_div(a.M, 16)), dim3(256), 0, s, a, reinterpret_cast<float*>(a.workspace));
    return launch(false);
  };
  switch (a.epilogue) {
    case UA2_EPI_QKV_ROPE: {
      const bool lm = a.rope_mode == UA2_ROPE_HALF_SPLIT && (a.kv.head_size == 128 || a.kv.head_size == 64) && !a.bias;
      const bool dit = a.rope_mode == UA2_ROPE_NONE && a.kv.head_size == 64;
      if (!(lm || dit) || !aligned16(a.q_out) || !aligned16(a.kv.k_pool) || !aligned16(a.kv.v_pool)) return 1;
      if (lm && (!aligned16(a.rope_cos) || !aligned16(a.rope_sin))) return 1;
      return with_rstd([&](bool dry) { return launch2<UA2_EPI_QKV_ROPE>(a, s, rstd, dry); });
    }
    case UA2_EPI_STORE:
      if (!a.y && !a.part_max) return 1;
      if (a.y && (a.ldy % 4 || !aligned16(a.y))) return 1;
      return with_rstd([&](bool dry) { return launch2<UA2_EPI_STORE>(a, s, rstd, dry); });
    case UA2_EPI_RESIDUAL:
      if (rstd) return 1;
      if (a.ldy % 4 || a.ldr % 4 || !aligned16(a.y) || !aligned16(a.resid) || (a.out_scale && !aligned16(a.out_scale))) return 1;
      return launch2<UA2_EPI_RESIDUAL>(a, s, nullptr, dry_run);
    case UA2_EPI_SWIGLU:
    case UA2_EPI_GELU:
      if (a.y && (a.ldy % 4 || !aligned16(a.y))) return 1;
      if (a.y_packed && !aligned16(a.y_packed)) return 1;
      return with_rstd([&](bool dry) { return glu ? launch2<UA2_EPI_SWIGLU>(a, s, rstd, dry) : launch2<UA2_EPI_GELU>(a, s, rstd, dry); });
    default: return 1;
  }
}

int ua2_gemm2_try_launch(const ua2_linear_args& a, hipStream_t s) { return gemm2_try(a, s, false); }

// ua2hip.h: would ua2_linear run this launch on the order-free kernel?  (The operand must be given packed or a workspace for the prep
// launch supplied, as for every many-row launch.)
extern "C" int ua2_linear_order_free_accepts(const ua2_linear_args* a) {
  if (!a || a->sum_order != UA2_SUM_ORDER_FREE || a->M <= 0 || a->N <= 0 || a->K <= 0) return 0;
  if (!a->x_packed && (!a->workspace || a->workspace_bytes < ua2_linear_workspace_bytes(a->dtype, a->M, a->K))) return 0;
  return gemm2_try(*a, nullptr, true) == 0 ? 1 : 0;
}
