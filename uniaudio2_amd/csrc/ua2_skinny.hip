// Batched-decode form of ua2_linear (a few dozen to a few hundred rows: 64 live sequences per GPU, SURVEY.md §8d
// config 4) — "weights stationary, operand streaming".
//
// Replaces the same reference code as ua2_gemv.hip (lit_model.py:382-511 qkv / proj, :591-595 LLaMAMLP; model_new.py:617-641
// heads) when 17..~512 rows share a launch.  Contract: bit-identical, row by row, with the decode kernel: K is split over
// `nw` contiguous chunk ranges exactly as ua2_gemv.hip splits it (one wave per range, an MFMA chain from zero each, partial
// sums added in range order), so a row's bits are those of the B = 1 run whatever this kernel's tile parameters are.
//
// What changed against round 2's skinny kernel (ua2_gemm.hip, kept for fp32 and for shapes outside the table below), and why
// (profiles/r3_notes.md):
//   * the old kernel walked a wave's range in rounds of 4 chunks — weights AND operand fragments requested, then waited
//     for, then multiplied — so every round exposed a full memory round trip (K = 8192: four of them), and at 64 rows the
//     operand is 4 KiB per 1 KiB of weights: the launches cost 6-12 us more than the B = 1 GEMV on the same weights;
//   * here a wave requests ALL the weight fragments of its range up front (non-temporal 1 KiB bursts straight to
//     registers, exactly the decode kernel's pattern: the whole matrix is in flight at once) and keeps them — the weights
//     are stationary; the packed operand fragments (L2-resident, written by the producer in fragment order) stream
//     through a small register ring `LA` chunks ahead of the MFMAs;
//   * a workgroup may take several PASSES of MT row tiles with the same resident weights (B = 256: the weights cross HBM
//     and L2 once per 16 x MT x passes rows instead of once per 64), the ring running across the pass boundary;
//   * CT column tiles per wave share each operand fragment (halves the L2 -> CU operand traffic where N is large enough
//     that halving the grid still fills the chip).
// Bound: HBM on the weights (the operand comes from L2: 64 rows x K x 2 B per workgroup — at 56 B/clk/CU that is 2.5 us at
// K = 3072 and 6.8 us at K = 8192 per workgroup, in the shadow of the weight stream).
#include <stdlib.h>

#include <algorithm>

#include "ua2_common.h"
#include "ua2_linear_common.h"

namespace {

constexpr int kRsrcFlags = 0x00020000;   // raw buffer, dword data format (gfx9 family)

// SC: the launch is a UA2_PRO_SCALED consumer (row scales from the producer's partials) — a template flag, not a runtime
// test: the partials' registers must not exist in the other instantiations (as a runtime branch they spilled all of them)
// WD (round 6): 0 = the wave's weight fragments all resident (above); WD > 0 = a register RING of WD chunks per stream, refilled as the
// chunks are consumed — one pass only (nothing is re-used), which is the 17-64-row case.  It trades nothing in bytes in flight when CT
// doubles with it (SwiGLU at K = 3072: 2 matrices x 2 column tiles x 6 chunks = the 24 KiB per wave of the resident CT = 1 form), and
// it is what lets SwiGLU take two column tiles per wave at all: resident, 2 x 2 x 12 fragments are 192 of a wave's 256 registers.
// Refills go out BEHIND the operand refill of the same chunk (a wave's loads retire in order: the operand is needed next chunk, the
// weights WD chunks on).
// RPW (round 6): K ranges per wave.  NWV stays the number of RANGES (the bit contract: `red` holds one partial per range, added in range
// order); with RPW = 2 a workgroup is NWV / 2 waves that walk two consecutive ranges each — all 2 x CH weight fragments in flight at once as
// before, and twice the registers per wave for a deeper operand ring (K = 8192 at 16 ranges: 128 registers per wave leave LA = 1).
template <int EPI, int CT, int MT, int CH, int NWV, int LA, bool SC, int WD = 0, int RPW = 1>
__global__ __launch_bounds__(NWV / RPW * 64) void skinny2_kernel(const ua2_linear_args a, const u32x4* __restrict__ apack, const int passes) {
  constexpr int DT = UA2_BF16;
  constexpr int NM = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;   // weight matrices
  constexpr int NS = NM * CT;                           // weight streams per wave: stream s = matrix s / CT, column tile s % CT
  constexpr int CHT = RPW * CH;                         // chunks a wave walks
  constexpr int NWAVES = NWV / RPW;
  static_assert(NWV % RPW == 0 && NWAVES % 4 == 0 && (RPW == 1 || WD == 0), "whole thread groups; the weight ring is a one-range form");
  static_assert(CHT % LA == 0 && LA <= CHT, "the operand ring must tile a wave's chunks");
  constexpr int WR = WD > 0 ? WD : CHT;                 // weight chunks a wave holds at a time
  static_assert(WR <= CHT && CHT % WR == 0, "the weight ring must tile a range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);          // [NWV][NS][MT][256]
  float* rstd_l = red + NWV * NS * MT * 256;            // [MT * passes * 16] row scales (UA2_PRO_SCALED)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int nchunks = NWV * CH;                     // checked by the launcher: K / KC == NWV * CH
  const int mtiles = (a.M + 15) / 16, ntiles = (a.N + 15) / 16;
  const int c0 = wave * CHT;

  // Addressing: buffer loads — a scalar resource + a scalar offset (tile, chunk) + ONE per-lane 32-bit offset, so a load
  // costs no 64-bit vector address arithmetic and no address registers (with flat global loads the compiler kept a 64-bit
  // VGPR pair per stream and spilled operand fragments at 16 waves per workgroup).
  const unsigned voff = (unsigned)(c0 * 64 + lane) * 16u;
  __amdgpu_buffer_rsrc_t wr[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int nt = min((int)blockIdx.x * CT + s % CT, ntiles - 1);
    const char* base = reinterpret_cast<const char*>((s / CT) ? a.w1 : a.w0) + (size_t)nt * nchunks * 1024;
    wr[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nchunks * 1024, kRsrcFlags);
  }
  // the operand: [mtiles][nchunks] fragments of 1 KiB (< 4 GiB: the launcher checks)
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(apack), 0, (unsigned)mtiles * (unsigned)(nchunks * 1024), kRsrcFlags);
  const int mt_first = blockIdx.y * passes * MT;
  // byte offset of row tile mt's fragments; clamped: a tile past M is read (valid memory) and dropped
  auto aoff = [&](int mt) { return (unsigned)min(mt, mtiles - 1) * (unsigned)(nchunks * 1024); };
  auto lda = [&](unsigned tile_off, int chunk) { return __builtin_amdgcn_raw_buffer_load_b128(ar, voff, tile_off + (unsigned)chunk * 1024u, 0); };

  auto ldw = [&](int s, int chunk) { return __builtin_amdgcn_raw_buffer_load_b128(wr[s], voff, chunk * 1024, 2); };   // aux 2 = nt: streamed once

  // UA2_PRO_SCALED: sum-of-squares partials of the pass's rows — a 16-lane group per row, SW sweeps over the MT * 16 rows —
  // requested at the head of the pass (pass 0: before the weight burst), reduced after its chunk loop, into rstd_l
  constexpr int SW = SC ? (MT * 16 + NWAVES * 4 - 1) / (NWAVES * 4) : 1;
  constexpr int NPL = SC ? (nchunks + 7) / 8 : 1;         // K / 16 partials per row over 16 lanes
  float ssqv[SW][NPL];
  auto ssq_request = [&](int pass) {
#pragma unroll
    for (int w = 0; w < SW; ++w) scaled_ssq_request(a, (mt_first + pass * MT) * 16 + w * NWAVES * 4 + (tid >> 4), tid & 15, ssqv[w]);
  };
  auto ssq_reduce = [&](int pass) {
#pragma unroll
    for (int w = 0; w < SW; ++w) {
      const int r = w * NWAVES * 4 + (tid >> 4);
      const float rs = scaled_rstd_reduce(a, tid & 15, ssqv[w]);
      if ((tid & 15) == 0 && r < MT * 16) rstd_l[pass * MT * 16 + r] = rs;
    }
  };
  if constexpr (SC) ssq_request(0);

  // Issue order: a wave's loads retire in order, so the first operand chunks (L2 hits, needed first) go out BEFORE the weight
  // burst (HBM).  (A "rolling window" form — weights and operand of a chunk travelling together through a ring, consumed in
  // issue order — was measured and lost 5-20 %: it caps the weight bytes in flight per wave at the ring depth,
  // profiles/r3_skinny_sweep.txt.)
  u32x4 af[LA][MT], wf[NS][WR];
  {
    unsigned o[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) o[mi] = aoff(mt_first + mi);
#pragma unroll
    for (int u = 0; u < LA; ++u)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) af[u][mi] = lda(o[mi], u);
  }
  __builtin_amdgcn_sched_barrier(0);                    // the machine scheduler may not hoist the weight burst above these
#pragma unroll
  for (int u = 0; u < WR; ++u)
#pragma unroll
    for (int s = 0; s < NS; ++s) wf[s][u] = ldw(s, u);
  __builtin_amdgcn_sched_barrier(0);
  // the partials were requested first (they retire first, ~an L2 round trip): reduced here, while the weights stream, so
  // that their registers are dead before the chunk loop (kept across it, the allocator starved the loop's operand ring of
  // lookahead: vmcnt(1) instead of vmcnt(3) per refill, +3 us per pass)
  if constexpr (SC) ssq_reduce(0);

  for (int pass = 0;;) {                                 // the first pass is unconditional (the launcher never starts a workgroup past M):
    const int mtp = mt_first + pass * MT;               // a guard in front of it lets the compiler sink the first operand loads below the burst
    unsigned oc[MT], on[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) { oc[mi] = aoff(mtp + mi); on[mi] = aoff(mtp + MT + mi); }
    f32x4 acc[NS][MT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[s][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CHT; ++u) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        AFrag<DT> f;
        f.v = af[u % LA][mi];
#pragma unroll
        for (int s = 0; s < NS; ++s) f.mma(wf[s][u % WR], acc[s][mi]);
      }
      // refill the slot just consumed: chunk u + LA of this pass, or the first chunks of the next pass (always requested —
      // a branch around a load costs a full vmcnt(0) somewhere; past the last pass the clamped tile is read and dropped)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        af[u % LA][mi] = (u + LA < CHT) ? lda(oc[mi], u + LA) : lda(on[mi], u + LA - CHT);
      if constexpr (WD > 0) {                            // the weight ring: chunk u's slot takes chunk u + WD (behind the operand refill)
        if (u + WR < CHT) {
#pragma unroll
          for (int s = 0; s < NS; ++s) wf[s][u % WR] = ldw(s, u + WR);
        }
      }
      if (u % CH == CH - 1) {                            // end of a range: its partial chain goes to the range's slot of `red`
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            *reinterpret_cast<f32x4*>(&red[((((wave * RPW + u / CH) * NS + s) * MT) + mi) * 256 + lane * 4]) = acc[s][mi];
            acc[s][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
    // Epilogue: the (column tile, row tile) items of the pass are dealt round-robin to the workgroup's NG = NWV / 4 thread groups of 256
    // (round 6: before, threads 0-255 walked all of them while the other waves idled at the barrier — at 64 rows that is 4-8
    // items of LDS reduction + stores in a row).  Every item's epilogue loads (residual values; position -> RoPE table entry /
    // page id; norm weight) go out one item ahead — the first before the barrier: issued where they are needed they are a dependent L2 round
    // trip or two in front of the stores (all items ahead at once cost 8-140 B of scratch in the multi-pass forms, whose weights stay resident).
#ifdef UA2_SKINNY_ONE_GROUP                              // A/B build: the round-3 form (threads 0-255 walk every item)
    constexpr int NG = 1, NI = CT * MT, IPG = NI;
#else
    constexpr int NG = NWAVES / 4, NI = CT * MT, IPG = (NI + NG - 1) / NG;
#endif
    // a thread's element of a 16 x 16 tile, from an OPAQUE copy of its id: derived from `tid` itself the compiler hoists the epilogue's
    // per-thread addresses (64-bit, loop-invariant over the passes) above the chunk loop and spills them across it
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int grp = tl >> 8, row = (tl & 255) >> 4, col = tl & 15;
    const int srcl = (((row >> 2) << 4) + col) * 4 + (row & 3);
    auto prefetch = [&](int it, EpiPre& p) {             // both stages of item `it` (a no-op past the last item)
      if (it >= NI || grp >= NG) return;
      const int nt = min((int)blockIdx.x * CT + it / MT, ntiles - 1), m0 = (mtp + it % MT) * 16;
      epilogue_prefetch_a<DT, EPI>(a, nt, row, col, p, m0);
      epilogue_prefetch_b<DT, EPI>(a, nt, row, col, p, m0);
    };
    EpiPre pre;
    prefetch(grp, pre);
    ua2_lds_barrier();                                   // LDS-only hand-off: the operand prefetch of the next pass stays in flight
    const bool more_passes = pass + 1 < passes && mt_first + (pass + 1) * MT < mtiles;    // uniform
    if constexpr (SC) { if (more_passes) ssq_request(pass + 1); }                         // lands under the epilogue below
#pragma unroll
    for (int k = 0; k < IPG; ++k) {
      const int it = grp + k * NG;                        // uniform over the group's four waves
      EpiPre nxt;                                        // the next item's loads travel under this item's reduction and stores
      if (k + 1 < IPG) prefetch(it + NG, nxt);
      const int ct = it / MT, mi = it % MT;
      const int nt = blockIdx.x * CT + ct, m0 = (mtp + mi) * 16;
      if (it < NI && grp < NG && nt < ntiles && m0 < a.M) {
        int tile[NM];
#pragma unroll
        for (int t = 0; t < NM; ++t) tile[t] = nt;
        if constexpr (SC) pre.rstd = rstd_l[min((mtp - mt_first + mi) * 16 + row, passes * MT * 16 - 1)];
        float v[NM];
#pragma unroll
        for (int t = 0; t < NM; ++t) {
          float sacc = 0.f;
#pragma unroll
          for (int w = 0; w < NWV; ++w) sacc += red[(((w * NS + t * CT + ct) * MT) + mi) * 256 + srcl];
          v[t] = sacc;
        }
        linear_epilogue<DT, EPI, NM>(a, v, tile, row, col, pre, m0, min(16, a.M - m0));
      }
      if (k + 1 < IPG) pre = nxt;
    }
    if (!more_passes) break;
    ++pass;
    if constexpr (SC) ssq_reduce(pass);
    ua2_lds_barrier();                                   // `red` is rewritten by the next pass
  }
}

// ---- K ranges across WORKGROUPS (round 6; the "wide-column K split + combine" of profiles/r4_notes.md §6) -----------------------------
// At 33-64 rows the K = 8192 down-projection is bounded by the operand every 16-column workgroup pulls in again (192 x 1.26 MB = 242 MB
// through L2 for 50 MB of weights; ring depth moves nothing, r6 notes §10).  Here a workgroup owns ONE of the NWV ranges and a wide group of
// column tiles: its 4 waves share the range's operand (MT x CH fragments, staged once in LDS: 64 KiB at 64 rows) and hold CT column
// tiles' weights each, all in flight at once — 256 workgroups x (64 KiB operand + 48 CT KiB weights).  Each wave's MFMA chain over the
// range's chunks is the chain wave `range` of skinny2_kernel / gemv_kernel runs (same fragments, same order, from zero); the partial
// tiles go to range_ws as [range][row tile][column tile][256 floats] in accumulator order, and rsplit_combine_kernel adds them in range
// order from 0.f — `sacc += red[w]` of the kernels above, word for word — and runs the same linear_epilogue.  Same bits
// (tests/test_gpu_invariance.py::test_range_split_...).
// MEASURED (profiles/r6_range_split.txt): main 15.3 us (trunk, 3072 columns) / 13.3 (depth decoder, 2048) + combine 5.9 against 20.4 / 14.3 us for
// skinny2_kernel — the B = 64 frame 4.93 -> 5.09 ms.  The main launch is a weight burst (6 us of HBM + a round trip), then the staged
// operand's barrier, then 192 MFMAs per wave on one wave per SIMD, then 12 KiB of partial stores per wave: phases in a row, and the 12.6 MB
// of partials cost a second launch.  So the form is OPT-IN: an op-level caller that hands range_ws gets it; the frame executor does not
// (UA2_RANGE_SPLIT=1 makes it, for A/B).
template <int CT, int MT, int CH>
__global__ __launch_bounds__(256, 1) void rsplit_main_kernel(const ua2_linear_args a, const u32x4* __restrict__ apack, float* __restrict__ part, const int nranges) {
  constexpr int DT = UA2_BF16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* alds = reinterpret_cast<u32x4*>(smem);         // [MT][CH][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int range = blockIdx.y, nchunks = nranges * CH;
  const int mtiles = (a.M + 15) / 16, ntiles = (a.N + 15) / 16;
  const int c0 = range * CH;
  const unsigned voff = (unsigned)(c0 * 64 + lane) * 16u;
  // weights first (HBM: the long round trip), non-temporal
  u32x4 wf[CT][CH];
  int nt[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    nt[c] = ((int)blockIdx.x * 4 + wave) * CT + c;
    const char* base = reinterpret_cast<const char*>(a.w0) + (size_t)min(nt[c], ntiles - 1) * nchunks * 1024;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nchunks * 1024, kRsrcFlags);
#pragma unroll
    for (int u = 0; u < CH; ++u) wf[c][u] = __builtin_amdgcn_raw_buffer_load_b128(wr, voff, u * 1024, 2);
  }
  // the range's operand, once per workgroup: MT x CH fragments over the 4 waves, through registers into LDS
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(apack), 0, (unsigned)mtiles * (unsigned)(nchunks * 1024), kRsrcFlags);
  constexpr int NF = MT * CH, FPW = (NF + 3) / 4;
  u32x4 stg[FPW];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int f = min(wave + 4 * k, NF - 1), mi = f / CH, u = f - mi * CH;
    stg[k] = __builtin_amdgcn_raw_buffer_load_b128(ar, voff, (unsigned)min(mi, mtiles - 1) * (unsigned)(nchunks * 1024) + (unsigned)u * 1024u, 0);
  }
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int f = wave + 4 * k;
    if (f < NF) alds[f * 64 + lane] = stg[k];
  }
  __syncthreads();
  f32x4 acc[CT][MT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[c][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CH; ++u)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      AFrag<DT> f;
      f.v = alds[(mi * CH + u) * 64 + lane];
#pragma unroll
      for (int c = 0; c < CT; ++c) f.mma(wf[c][u], acc[c][mi]);
    }
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if (nt[c] >= ntiles) continue;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      if (mi >= mtiles) break;
      *reinterpret_cast<f32x4*>(part + ((((size_t)range * mtiles + mi) * ntiles + nt[c]) * 256 + lane * 4)) = acc[c][mi];
    }
  }
}

template <int EPI, int NR>
__global__ __launch_bounds__(256) void rsplit_combine_kernel(const ua2_linear_args a, const float* __restrict__ part) {
  constexpr int DT = UA2_BF16;
  const int mtiles = (a.M + 15) / 16, ntiles = (a.N + 15) / 16;
  const int nt = blockIdx.x % ntiles, mi = blockIdx.x / ntiles;
  const int tid = threadIdx.x, row = tid >> 4, col = tid & 15;
  const int srcl = (((row >> 2) << 4) + col) * 4 + (row & 3);
  const int m0 = mi * 16;
  EpiPre pre;
  epilogue_prefetch<DT, EPI>(a, nt, row, col, pre, m0);
  const float* p = part + (((size_t)mi * ntiles + nt) * 256 + srcl);
  const size_t rstride = (size_t)mtiles * ntiles * 256;
  float t[NR];                                           // all NR loads in flight, then the sum in range order from zero
#pragma unroll
  for (int w = 0; w < NR; ++w) t[w] = p[(size_t)w * rstride];
  float v[1];
  float sacc = 0.f;
#pragma unroll
  for (int w = 0; w < NR; ++w) sacc += t[w];
  v[0] = sacc;
  int tile[1] = {nt};
  linear_epilogue<DT, EPI, 1>(a, v, tile, row, col, pre, m0, min(16, a.M - m0));
}

// Returns 0 when launched, 1 when the problem is outside this form.
int rsplit_try_launch(const ua2_linear_args& a, ua2_gemv_geometry geo, hipStream_t s) {
  if (!a.range_ws || a.dtype != UA2_BF16 || a.epilogue != UA2_EPI_RESIDUAL || a.prologue == UA2_PRO_SCALED || a.K % 32) return 1;
  const int nchunks = a.K / 32, mtiles = ua2_ceil_div(a.M, 16), ntiles = ua2_ceil_div(a.N, 16);
  if (geo.waves != 16 || nchunks != 16 * 16 || mtiles < 3 || mtiles > 4 || a.N % 16) return 1;       // K = 8192, 33-64 rows
  const int ct = ntiles % 12 == 0 && ntiles / 12 >= 12 ? 3 : (ntiles % 8 == 0 ? 2 : 0);               // 4 waves x CT tiles per workgroup, >= 192 workgroups with 3
  if (!ct) return 1;
  const size_t need = (size_t)geo.waves * mtiles * ntiles * 256 * sizeof(float);
  if (a.range_ws_bytes < need || (reinterpret_cast<uintptr_t>(a.range_ws) & 15)) return 1;
  const u32x4* apack = reinterpret_cast<const u32x4*>(a.x_packed ? a.x_packed : a.workspace);
  const dim3 grid(ntiles / (4 * ct), geo.waves);
  constexpr size_t smem = 4 * 16 * 1024;
  if (ct == 3) {
    constexpr auto kern = rsplit_main_kernel<3, 4, 16>;
    ua2_allow_big_lds<kern>();
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a, apack, a.range_ws, geo.waves);
  } else {
    constexpr auto kern = rsplit_main_kernel<2, 4, 16>;
    ua2_allow_big_lds<kern>();
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a, apack, a.range_ws, geo.waves);
  }
  hipLaunchKernelGGL((rsplit_combine_kernel<UA2_EPI_RESIDUAL, 16>), dim3(ntiles * mtiles), dim3(256), 0, s, a, a.range_ws);
  ua2_count_launch(UA2_CNT_RSPLIT);
  UA2_LAUNCH_CHECK();
  return 0;
}

// VGPRs a variant needs: resident weights + operand ring + accumulators + addressing / epilogue slack.  Variants over the
// per-wave budget (512 per SIMD shared by NWV / 4 waves) spill and are not built.
constexpr int regs_needed(int nm, int ct, int mt, int ch, int la, int nwv = 16, bool sc = false, int wd = 0, int rpw = 1) {
  const int sw = (mt * 16 + nwv / rpw * 4 - 1) / (nwv / rpw * 4), npl = (nwv * ch + 7) / 8;      // the scaled consumer's sum-of-squares partials
  // ring forms: the partials are dead before the accumulators come alive (reduced behind the burst, in front of the chunk loop): only
  // what they need beyond the accumulators' registers counts
  const int acc = nm * ct * mt * 4, scx = sc ? sw * npl + 40 : 0;
  return nm * ct * (wd > 0 ? wd : ch * rpw) * 4 + la * mt * 4 + acc + 20 + (wd > 0 ? (scx > acc ? scx - acc : 0) : scx);
}

struct Variant { int ct, mt, la, passes, wd = 0, rpw = 1; };

// experiment hook: UA2_SKINNY2="ct,mt,la,passes" (read per call); "off" disables the kernel
bool env_variant(Variant& v, bool& off) {
  const char* e = getenv("UA2_SKINNY2");
  off = false;
  if (!e) return false;
  if (e[0] == 'o') { off = true; return false; }
  v.wd = 0; v.rpw = 1;
  return sscanf(e, "%d,%d,%d,%d,%d,%d", &v.ct, &v.mt, &v.la, &v.passes, &v.wd, &v.rpw) >= 4;
}

template <int EPI, int CT, int MT, int CH, int NWV, int LA, bool SC, int WD = 0, int RPW = 1>
int launch_one(const ua2_linear_args& a, int passes, hipStream_t s) {
  constexpr int NM = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr auto kern = skinny2_kernel<EPI, CT, MT, CH, NWV, LA, SC, WD, RPW>;
  constexpr size_t red_bytes = (size_t)NWV * NM * CT * MT * 1024;
  if constexpr (red_bytes > 156 * 1024) return 1;
  const size_t smem = red_bytes + (size_t)passes * MT * 16 * sizeof(float);
  ua2_allow_big_lds<kern>();
  const int mtiles = ua2_ceil_div(a.M, 16), ntiles = ua2_ceil_div(a.N, 16);
  const dim3 grid(ua2_ceil_div(ntiles, CT), ua2_ceil_div(mtiles, MT * passes));
  hipLaunchKernelGGL(kern, grid, dim3(NWV / RPW * 64), smem, s, a, reinterpret_cast<const u32x4*>(a.x_packed ? a.x_packed : a.workspace), passes);
  ua2_count_launch(UA2_CNT_SKINNY2);
  return 0;
}

template <int EPI, int CH, int NWV>
int launch_variant(const ua2_linear_args& a, const Variant& v, hipStream_t s) {
  constexpr int NM = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
#define UA2_SK(CT_, MT_, LA_)                                                                            \
  if (v.ct == CT_ && v.mt == MT_ && v.la == LA_) {                                                       \
    if constexpr (EPI != UA2_EPI_RESIDUAL && CH % LA_ == 0 && regs_needed(NM, CT_, MT_, CH, LA_, NWV, true) <= 2048 / NWV) { \
      if (a.prologue == UA2_PRO_SCALED) return launch_one<EPI, CT_, MT_, CH, NWV, LA_, true>(a, v.passes, s);             \
    }                                                                                                    \
    if (a.prologue == UA2_PRO_SCALED) return 1;                                                          \
    if constexpr (CH % LA_ == 0 && regs_needed(NM, CT_, MT_, CH, LA_) <= 2048 / NWV) \
      return launch_one<EPI, CT_, MT_, CH, NWV, LA_, false>(a, v.passes, s);                              \
    return 1;                                                                                            \
  }
  // weight-ring forms (one pass): the ring buys the registers for more column tiles per wave than fit resident — fewer operand
  // bytes into the CU per weight byte, which is what a 33-64-row launch waits for
#define UA2_SKR(CT_, MT_, LA_, WD_)                                                                                              \
  if (v.wd == WD_ && v.passes == 1 && v.ct == CT_ && v.mt == MT_ && v.la == LA_) {                                                 \
    if constexpr (EPI != UA2_EPI_RESIDUAL && WD_ < CH && CH % WD_ == 0 && CH % LA_ == 0 &&                                         \
                  regs_needed(NM, CT_, MT_, CH, LA_, NWV, true, WD_) <= 2048 / NWV) {                                              \
      if (a.prologue == UA2_PRO_SCALED) return launch_one<EPI, CT_, MT_, CH, NWV, LA_, true, WD_>(a, 1, s);                        \
    }                                                                                                                            \
    if (a.prologue == UA2_PRO_SCALED) return 1;                                                                                  \
    if constexpr (WD_ < CH && CH % WD_ == 0 && CH % LA_ == 0 && regs_needed(NM, CT_, MT_, CH, LA_, NWV, false, WD_) <= 2048 / NWV)     \
      return launch_one<EPI, CT_, MT_, CH, NWV, LA_, false, WD_>(a, 1, s);                                                         \
    return 1;                                                                                                                    \
  }
  if constexpr (EPI == UA2_EPI_SWIGLU) {
    UA2_SKR(2, 4, 1, 6) UA2_SKR(2, 4, 2, 6) UA2_SKR(2, 4, 1, 4) UA2_SKR(2, 4, 1, 3)
  }
#undef UA2_SKR
  if (v.wd != 0) return 1;
  // two ranges per wave (K = 8192: 16 ranges of 16 chunks on 8 waves; RESIDUAL has no scaled consumer form)
  if constexpr (EPI == UA2_EPI_RESIDUAL && CH == 16 && NWV == 16) {
#define UA2_SK2(MT_, LA_)                                                                                   \
    if (v.rpw == 2 && v.ct == 1 && v.mt == MT_ && v.la == LA_) {                                            \
      if (a.prologue == UA2_PRO_SCALED) return 1;                                                           \
      if constexpr (regs_needed(NM, 1, MT_, CH, LA_, NWV, false, 0, 2) <= 2048 / (NWV / 2))                 \
        return launch_one<EPI, 1, MT_, CH, NWV, LA_, false, 0, 2>(a, v.passes, s);                          \
      return 1;                                                                                             \
    }
    UA2_SK2(4, 4) UA2_SK2(2, 4) UA2_SK2(4, 1)          // experiment forms (tools/ubench/skinny_shapes.py)
#undef UA2_SK2
  }
  if (v.rpw != 1) return 1;
  UA2_SK(1, 4, 1) UA2_SK(1, 4, 2) UA2_SK(1, 2, 2)
  UA2_SK(2, 4, 1) UA2_SK(2, 4, 2) UA2_SK(2, 2, 2)
#undef UA2_SK
  return 1;
}

// Tile parameters from the shape: measured, tools/ubench/skinny_shapes.py (profiles/r3_skinny_sweep.txt).  All variants give
// the same bits, so this is purely a cost choice.  What the sweep says: the launch is bound by bytes INTO each CU (weights +
// 64 rows x K of operand per workgroup, ~60-75 GB/s per CU achieved), so (i) two column tiles per wave wherever the weights
// fit the registers and the grid stays >= ~96 column groups (halves the operand bytes per weight byte); (ii) when there are
// fewer column groups than CUs, the rows are split over 256 / groups workgroups (the weights cross L2 twice, the operand
// ingest per CU halves); (iii) otherwise one workgroup walks all rows in passes with its weights resident.
Variant pick_variant(const ua2_linear_args& a, int waves, int ch, int nm) {
  const int mtiles = ua2_ceil_div(a.M, 16), ntiles = ua2_ceil_div(a.N, 16);
  const int budget = 2048 / waves;
  const bool sc = a.prologue == UA2_PRO_SCALED;
  Variant v{1, 4, 1, 1};
  if (ntiles >= 192 && regs_needed(nm, 2, 2, ch, 2, waves, sc) <= budget) v.ct = 2;
  const int groups = ua2_ceil_div(ntiles, v.ct);
  const int ysplit = std::max(1, std::min(256 / groups, ua2_ceil_div(mtiles, 2)));
  const int rows = ua2_ceil_div(mtiles, ysplit);            // row tiles per workgroup
  v.mt = (rows >= 4 && regs_needed(nm, v.ct, 4, ch, 1, waves, sc) <= budget) ? 4 : 2;
  v.la = (v.mt == 2 && ch % 2 == 0) ? 2 : 1;
  v.passes = ua2_ceil_div(rows, v.mt);
  // 33-64 rows of a SwiGLU launch (round 6, profiles/r6_notes.md §10): one pass of four row tiles, so nothing re-uses the weights and a
  // register RING of them frees the registers for two column tiles per wave — the operand crosses into half as many workgroups
  // (trunk, 64 rows: 30.2 -> 23.9 us in the frame with a four-chunk ring).  Ring = half of the range where it fits, else four chunks.
  static const char* ring_env = getenv("UA2_SKINNY_RING");                 // A/B (read once): "off" or "wd,la"
  if (!(ring_env && ring_env[0] == 'o') && nm == 2 && mtiles >= 3 && mtiles <= 4 && ntiles >= 192) {
    int wd = 0, la = 1;
    if (ring_env && sscanf(ring_env, "%d,%d", &wd, &la) == 2) {
    } else if (ch == 12 && regs_needed(nm, 2, 4, ch, 2, waves, sc, 6) <= budget) { wd = 6; la = 2; }
    else if (ch == 12 && regs_needed(nm, 2, 4, ch, 1, waves, sc, 6) <= budget) { wd = 6; la = 1; }
    else if (ch % 4 == 0 && ch > 4 && regs_needed(nm, 2, 4, ch, 1, waves, sc, 4) <= budget) { wd = 4; la = 1; }
    if (wd > 0 && wd < ch && ch % wd == 0) v = Variant{2, 4, la, 1, wd};
  }
  // (Two K ranges per wave — UA2_SKINNY2="1,2,4,p,0,2", K = 8192 on eight waves with a four-chunk operand ring — measured 5 % ahead in
  // isolation, profiles/r6_skinny_rpw_sweep.txt, and level in the frame: 5.05 / 5.07 vs 5.07 / 5.08 ms at 64 rows.  Not picked.  What the
  // sweep settles: ring depths 1 / 2 / 4 / 8 are within 0.5 us of each other — the launch does not wait for a per-wave chain of round trips.)
  return v;
}

}  // namespace

// Returns 0 when launched, 1 when this problem is outside the kernel's table (the caller uses the older skinny kernel).
int ua2_skinny2_try_launch(const ua2_linear_args& a, ua2_gemv_geometry geo, hipStream_t s) {
  if (a.dtype != UA2_BF16) return 1;
  if (a.K % 32) return 1;
  const int nchunks = a.K / 32;
  if (nchunks % geo.waves) return 1;
  // the packed operand is addressed through one buffer resource (32-bit size / offsets): past 4 GiB the loads would wrap
  // and return zeros — leave such a problem to the tiled kernel
  if ((uint64_t)ua2_ceil_div(a.M, 16) * (uint64_t)nchunks * 1024ull >= (1ull << 32)) return 1;
  if (getenv("UA2_SKINNY2") == nullptr)                   // a forced variant means the sweep tool is measuring skinny2_kernel itself
    if (const int rc = rsplit_try_launch(a, geo, s); rc <= 0) return rc;
  const int ch = nchunks / geo.waves;
  const int nm = a.epilogue == UA2_EPI_SWIGLU ? 2 : 1;
  Variant v = pick_variant(a, geo.waves, ch, nm);
  bool off = false;
  Variant ev;
  const bool forced = env_variant(ev, off);
  if (forced) v = ev;
  if (off) return 1;
  if (v.passes < 1) v.passes = 1;
  int rc = 1;
#define UA2_GEO(EPI_, CH_, NWV_) \
  if (a.epilogue == EPI_ && ch == CH_ && geo.waves == NWV_) rc = launch_variant<EPI_, CH_, NWV_>(a, v, s);
  // K = 3072: 12 ranges x 8 chunks (grids <= 320 tiles) or 8 x 12 (large grids: SwiGLU, lm_head)
  UA2_GEO(UA2_EPI_QKV_ROPE, 8, 12) UA2_GEO(UA2_EPI_RESIDUAL, 8, 12) UA2_GEO(UA2_EPI_STORE, 8, 12)
  UA2_GEO(UA2_EPI_SWIGLU, 12, 8) UA2_GEO(UA2_EPI_STORE, 12, 8) UA2_GEO(UA2_EPI_QKV_ROPE, 12, 8)
  // K = 8192: 16 x 16
  UA2_GEO(UA2_EPI_RESIDUAL, 16, 16)
  // K = 2048: 16 x 4 (small grids) or 8 x 8 (SwiGLU, audio_head)
  UA2_GEO(UA2_EPI_QKV_ROPE, 4, 16) UA2_GEO(UA2_EPI_RESIDUAL, 4, 16) UA2_GEO(UA2_EPI_STORE, 4, 16)
  UA2_GEO(UA2_EPI_SWIGLU, 8, 8) UA2_GEO(UA2_EPI_STORE, 8, 8)
#undef UA2_GEO
  UA2_CHECK(!(forced && rc == 1), "UA2_SKINNY2=%d,%d,%d,%d,%d,%d is not built for this geometry (epilogue %d, %d ranges x %d chunks)", v.ct, v.mt,
            v.la, v.passes, v.wd, v.rpw, a.epilogue, geo.waves, ch);
  if (rc == 0) UA2_LAUNCH_CHECK();
  return rc;
}
