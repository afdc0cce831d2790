// Large-M form of ua2_linear: prefill of long prompts and batched decode (SURVEY.md §8d configs 3-5:
// 32 x 196-row prefill, 64 live sequences per GPU).
//
// Replaces the same reference code as ua2_linear.hip / ua2_gemv.hip (lit_model.py:382-511 qkv/proj,
// :591-595 LLaMAMLP, :883-890 RMSNorm; model_new.py:617-641 heads) when many rows share a launch.
//
// Contract: bit-identical, row by row, with the decode-regime kernel (ua2_gemv.hip).  That kernel sums a
// row's dot product as `waves` partial MFMA chains over contiguous chunk ranges, added in wave order,
// with `waves` a function of (dtype, N, K) only.  Here one wave owns a 64 x 64 output patch and walks
// the whole of K; at each of those range boundaries it retires the running chain into a second
// accumulator set (total += chain; chain = 0), which reproduces the same sums in the same order without
// splitting K across waves.  The price is 2x accumulator registers (128 of the 512 VGPRs); the gain is a
// proper GEMM: every weight byte is read once per 128 rows instead of once per 16.
//
// Two launches:
//   prep:  one workgroup per row applies the prologue (cast | RMSNorm | LayerNorm) exactly as the
//          decode kernel stages a row — same thread partition, same fixed-order statistics — and
//          writes the operand rows in MFMA fragment order [M/16][K/KC][64 lanes][16 B] (the layout
//          ua2_pack_linear gives the weights), so both GEMM operands stream as 1 KiB fragment blocks.
//   gemm:  workgroup = 4 waves (2 x 2) = 128 rows x 128 columns (SwiGLU: 64 columns of each matrix); K advances through an LDS
//          ring of fragment blocks — filled by LDS-DMA (global_load_lds, 1 KiB per wave-instruction, hand-counted vmcnt in
//          front of a raw s_barrier) on the 128- and 32-row tiles, through registers on the 64-row tile; every wave reads its
//          4 + 4 fragments per chunk with conflict-free 16-byte reads — the NEXT chunk's while it multiplies this one on the
//          128-row tile — and issues 16 MFMAs per chunk.  Workgroup ids are remapped so that each XCD works on a compact patch
//          of (row-block, column-block) pairs: the fragments an L2 fetches are reused by its 32 CUs.
// Bound: MFMA (2*M*N*K flop over (M + N)*K operand bytes); see DESIGN.md §5 for the measured fraction.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "ua2_common.h"
#include "ua2_linear_common.h"

namespace {

// ---- prep: prologue + fragment-order packing of the activation rows ---------------------------------
template <int DT, int PRO>
__global__ __launch_bounds__(1024) void gemm_prep_kernel(const ua2_linear_args a, void* __restrict__ apack) {
  constexpr int KC = Elem<DT>::KC, EPL = Elem<DT>::EPL, BYTES = Elem<DT>::BYTES;
  __shared__ float ssq[16], ssum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nw = nthreads >> 6;
  const int m = blockIdx.x;                      // grid covers ceil(M/16)*16 rows; rows >= M are zero-filled
  const int nchunks = (a.K + KC - 1) / KC;
  const bool live = m < a.M;
  const float* xr = a.x + (size_t)(live ? m : 0) * a.ldx;
  const bool ln = (PRO == UA2_PRO_NORM) && a.norm_kind == UA2_NORM_LAYERNORM;
  NormStat st{0.f, 1.f};
  if constexpr (PRO == UA2_PRO_NORM) {
    // identical to the decode kernel's pass 1: thread t owns k = 4t, 4t + 4*nthreads, ...
    float ss = 0.f, sm = 0.f;
    for (int k = tid * 4; k < a.K; k += nthreads * 4) {
      const float4 t = *reinterpret_cast<const float4*>(xr + k);
      ss = sumsq4(ss, t);
      sm = sum4(sm, t);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ss += __shfl_xor(ss, o); sm += __shfl_xor(sm, o); }
    if (lane == 0) { ssq[wave] = ss; ssum[wave] = sm; }
    __syncthreads();
    float t = 0.f, u = 0.f;
    for (int w = 0; w < nw; ++w) { t += ssq[w]; u += ssum[w]; }
    st = norm_stat(a, u, t);
  }
  char* base = reinterpret_cast<char*>(apack) + (size_t)(m >> 4) * nchunks * 1024 + (size_t)(m & 15) * 16;
  for (int k = tid * 4; k < nchunks * KC; k += nthreads * 4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && k < a.K) {
      t = *reinterpret_cast<const float4*>(xr + k);
      if constexpr (PRO == UA2_PRO_NORM) {
        const float4 w = *reinterpret_cast<const float4*>(a.norm_w + k);
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ln) b = *reinterpret_cast<const float4*>(a.norm_b + k);
        t.x = norm_apply(a, t.x, w.x, b.x, st);
        t.y = norm_apply(a, t.y, w.y, b.y, st);
        t.z = norm_apply(a, t.z, w.z, b.z, st);
        t.w = norm_apply(a, t.w, w.w, b.w, st);
      }
    }
    // fragment address of element k of row m: chunk c, lane = g*16 + (m & 15), element e
    const int c = k / KC, r = k - c * KC, g = r / EPL, e = r - g * EPL;
    char* dst = base + (size_t)c * 1024 + (size_t)g * 256 + (size_t)e * BYTES;
    if constexpr (DT == UA2_BF16) {
      uint2 p;
      p.x = (unsigned)f2bf(t.x) | ((unsigned)f2bf(t.y) << 16);
      p.y = (unsigned)f2bf(t.z) | ((unsigned)f2bf(t.w) << 16);
      *reinterpret_cast<uint2*>(dst) = p;
    } else {
      *reinterpret_cast<float4*>(dst) = t;
    }
  }
}

// ---- prep, one row TILE per workgroup (K <= 4096 bf16 / 2048 fp32) -------------------------------------------------
// The per-row kernel above writes 16 bytes of every 1 KiB fragment block from 16 different workgroups (8-byte pieces, 256 bytes
// apart) and spends a workgroup barrier on a row's statistics: 7.75 us for the DiT's 1000 x 1536 LayerNorm rows, two of them per
// layer.  Here a workgroup owns the 16 rows of a fragment row-tile, one wave per row.  The wave holds its row in registers
// (piece p = columns 4 * lane + 256 * p), reduces the statistics in the decode kernel's order — that kernel's thread
// t = 64 * vw + lane owns pieces p = vw, vw + NVW, ..., so the wave keeps NVW partial chains per lane, butterflies each across the
// lanes and adds them in vw order: the same operations on the same operands, no LDS, no barrier — and parks the normalised row
// in an LDS image; then every wave writes whole fragment blocks, 1 KiB per wave-instruction.  Same bits as the per-row kernel.
// ROWS = 4 (launches of fewer than 128 row tiles: the 16-row form would leave most CUs idle): four rows per workgroup, no image —
// a lane stores its 8-byte (bf16) pieces straight into the fragment blocks, as the per-row kernel does — and the norm's weight /
// bias requested together with the row instead of behind its statistics (one memory trip less on the critical path).
template <int DT, int PRO, int NVW, int MAXV, int ROWS>
__global__ __launch_bounds__(64 * ROWS) void gemm_prep16_kernel(const ua2_linear_args a, void* __restrict__ apack) {
  constexpr int KC = Elem<DT>::KC, EPL = Elem<DT>::EPL, BYTES = Elem<DT>::BYTES;
  constexpr bool IMG = ROWS == 16;                      // whole fragment tile in the workgroup: coalesce through an LDS image
  constexpr bool PREW = (PRO == UA2_PRO_NORM) && !IMG && MAXV <= 8;
  extern __shared__ __attribute__((aligned(16))) char prep_img[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunks = (a.K + KC - 1) / KC, kp = nchunks * KC;
  const int rowbytes = kp * BYTES + 16;                 // + 16 B: rows start 4 banks apart
  const int m = blockIdx.x * ROWS + wave;
  const bool live = m < a.M;
  const float* xr = a.x + (size_t)(live ? m : 0) * a.ldx;
  const bool ln = (PRO == UA2_PRO_NORM) && a.norm_kind == UA2_NORM_LAYERNORM;
  float4 v[MAXV];
  float4 nwv[PREW ? MAXV : 1], nbv[PREW ? MAXV : 1];
#pragma unroll
  for (int p = 0; p < MAXV; ++p) {
    const int k = 4 * lane + 256 * p;
    v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && k < a.K) v[p] = *reinterpret_cast<const float4*>(xr + k);
  }
  if constexpr (PREW) {
#pragma unroll
    for (int p = 0; p < MAXV; ++p) {
      const int k = 4 * lane + 256 * p;
      nwv[p] = nbv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < a.K) {
        nwv[p] = *reinterpret_cast<const float4*>(a.norm_w + k);
        if (ln) nbv[p] = *reinterpret_cast<const float4*>(a.norm_b + k);
      }
    }
  }
  NormStat st{0.f, 1.f};
  if constexpr (PRO == UA2_PRO_NORM) {
    // chains 0 .. NVW-1: sums of squares, NVW .. 2 NVW - 1: sums
    float ch[2 * NVW];
#pragma unroll
    for (int w = 0; w < 2 * NVW; ++w) ch[w] = 0.f;
#pragma unroll
    for (int p = 0; p < MAXV; ++p)                       // ascending p inside a chain = the decode kernel's ascending k
      if (4 * lane + 256 * p < a.K) { ch[p % NVW] = sumsq4(ch[p % NVW], v[p]); ch[NVW + p % NVW] = sum4(ch[NVW + p % NVW], v[p]); }
    // The decode kernel butterflies every chain over its 64 lanes (s += shfl_xor(s, 32), 16, ..., 1): 12 shuffles per virtual wave.
    // Same adds with a sixth of the shuffles: at each level two chains share one exchange — the lanes whose bit is clear keep
    // chain A (own + partner's A), the others chain B — so the registers halve per level and a chain's total ends up in the lanes
    // its index selects (bit 0 -> lane bit 32, bit 1 -> 16, ...).  own + other, as the butterfly: the same bits.
    constexpr int NCH = 2 * NVW;
    int n = NCH;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const bool hi = (lane & o) != 0;
      const int pairs = n / 2;
#pragma unroll
      for (int i = 0; i < NCH / 2; ++i) {
        if (i < pairs) {
          const float send = hi ? ch[2 * i] : ch[2 * i + 1], keep = hi ? ch[2 * i + 1] : ch[2 * i];
          ch[i] = keep + __shfl_xor(send, o);
        }
      }
      if (n & 1) {                                       // the odd one out: a plain butterfly level
        const float last = ch[n - 1];
        ch[pairs] = last + __shfl_xor(last, o);
      }
      n = pairs + (n & 1);
    }
    // lane holding chain w's total after the six levels
    auto lane_of = [](int w) {
      int n2 = NCH, l = 0;
      for (int o = 32; o >= 1; o >>= 1) {
        const int pairs = n2 / 2;
        if (w < 2 * pairs) { if (w & 1) l |= o; w >>= 1; } else { w = pairs; }
        n2 = pairs + (n2 & 1);
      }
      return l;
    };
    float t = 0.f, u = 0.f;
#pragma unroll
    for (int w = 0; w < NVW; ++w) {
      t += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ch[0]), lane_of(w)));
      u += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ch[0]), lane_of(NVW + w)));
    }
    st = norm_stat(a, u, t);
  }
  // image row (16-row form) or this row's 16-byte slot of fragment block 0 (4-row form: element k of row m lives in block
  // k / KC at lane (k % KC) / EPL * 16 + (m & 15), as gemm_prep_kernel addresses it)
  char* row = IMG ? prep_img + (size_t)wave * rowbytes
                  : reinterpret_cast<char*>(apack) + (size_t)(m >> 4) * nchunks * 1024 + (size_t)(m & 15) * 16;
#pragma unroll
  for (int p = 0; p < MAXV; ++p) {
    const int k = 4 * lane + 256 * p;
    if (k >= kp) continue;
    float4 t = v[p];
    if constexpr (PRO == UA2_PRO_NORM) {
      if (live && k < a.K) {
        float4 w, b = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PREW) { w = nwv[p]; b = nbv[p]; }
        else {
          w = *reinterpret_cast<const float4*>(a.norm_w + k);
          if (ln) b = *reinterpret_cast<const float4*>(a.norm_b + k);
        }
        t.x = norm_apply(a, t.x, w.x, b.x, st);
        t.y = norm_apply(a, t.y, w.y, b.y, st);
        t.z = norm_apply(a, t.z, w.z, b.z, st);
        t.w = norm_apply(a, t.w, w.w, b.w, st);
      }
    }
    size_t off = (size_t)k * BYTES;
    if constexpr (!IMG) {
      const int c = k / KC, r = k - c * KC, g = r / EPL, e = r - g * EPL;
      off = (size_t)c * 1024 + (size_t)g * 256 + (size_t)e * BYTES;
    }
    if constexpr (DT == UA2_BF16) {
      uint2 pk;
      pk.x = (unsigned)f2bf(t.x) | ((unsigned)f2bf(t.y) << 16);
      pk.y = (unsigned)f2bf(t.z) | ((unsigned)f2bf(t.w) << 16);
      *reinterpret_cast<uint2*>(row + off) = pk;
    } else {
      *reinterpret_cast<float4*>(row + off) = t;
    }
  }
  if constexpr (!IMG) return;
  __syncthreads();
  // fragment block c of this row-tile = [64 lanes][16 B]: lane g * 16 + r holds columns c * KC + g * EPL .. + EPL of row r
  const int g = lane >> 4, r = lane & 15;
  u32x4* dst = reinterpret_cast<u32x4*>(apack) + (size_t)blockIdx.x * nchunks * 64 + lane;
  for (int c = wave; c < nchunks; c += 16)
    dst[(size_t)c * 64] = *reinterpret_cast<const u32x4*>(prep_img + (size_t)r * rowbytes + (size_t)(c * KC + g * EPL) * BYTES);
}

// ---- the GEMM ------------------------------------------------------------------------------------------
// BMT 16-row tiles per workgroup: 8 (128 x 128 tile, 64 x 64 per wave) or 4 (64 x 128, 32 x 64 per wave — for launches
// whose 128-row grid would leave most CUs idle: M ~ 1000 rows x N = 1536 is 96 workgroups on 256 CUs).  A row's bits do
// not depend on the tile (same chains, same retire points).
constexpr int kKS = 2;      // chunks per LDS stage
constexpr int kGroupM = 8;  // row-blocks per L2 patch

// Slots of the LDS-DMA ring, per row-tile count.  A workgroup keeps slots - 1 chunks in flight; with an L2 round trip of
// ~0.5 us under load the ring delivers (slots - 1) x TILES KiB per round trip, which is what bounds the small-M launches
// (time per chunk flat at ~0.16 us = one round trip / 3 with four slots: tools/ubench/gemm_shapes.py, profiles/r4_notes.md §9).
#ifndef UA2_GEMM_RING_2
#define UA2_GEMM_RING_2 4
#endif
#ifndef UA2_GEMM_RING_4
#define UA2_GEMM_RING_4 4
#endif
#ifndef UA2_GEMM_RING_8
#define UA2_GEMM_RING_8 4
#endif
// Timing-only knock-outs of the LDS-DMA main loop (tools/ubench/build_alt.sh ... -DUA2_GEMM_DBG=<bits>; wrong results):
// 1 no ring refills after the prologue, 2 no MFMAs, 4 no fragment reads, 8 no workgroup barrier per chunk, 16 no epilogue,
// 32 no main loop at all (prologue fill, then straight to the epilogue).
#ifndef UA2_GEMM_DBG
#define UA2_GEMM_DBG 0
#endif
#if UA2_GEMM_DBG & 8
#define UA2_GEMM_BAR "s_nop 0"
#else
#define UA2_GEMM_BAR "s_barrier"
#endif
constexpr int ring_slots(int bmt) { return bmt == 2 ? UA2_GEMM_RING_2 : (bmt == 4 ? UA2_GEMM_RING_4 : UA2_GEMM_RING_8); }

// GL: the operand ring is filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction straight from L2 into the
// ring) instead of global -> registers -> ds_write_b128: see the main loop.  !GL (register staging) serves the hand-over
// instantiations (HO) and the UA2_GEMM_NO_GLDS experiment hook.
template <int DT, int EPI, int kBMT, bool HO, bool GL>
__global__ __launch_bounds__(256, (EPI == UA2_EPI_QKV_ROPE && kBMT <= 4 && GL) ? 3 : 2) void gemm_kernel(const ua2_linear_args a, const u32x4* __restrict__ apack, const int nw,
                                                      const int mblocks, const int nblocks, const int group_m, const int flags) {
  constexpr int KC = Elem<DT>::KC;
  constexpr int NWV = 4;                    // waves: 2 down the rows x 2 across the columns
  constexpr int kWM = kBMT / (NWV / 2);     // row tiles per wave
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr int WN = 4 / NT;          // column tiles per wave, per matrix
  constexpr int BNT = 2 * WN;         // column tiles per workgroup, per matrix
  constexpr int TILES = kBMT + NT * BNT;            // fragment streams per chunk (16)
  constexpr int KS = GL ? 1 : kKS;          // chunks per ring slot
  constexpr int LOADS = (TILES * KS + NWV - 1) / NWV;   // 16-byte pieces per thread per stage; 10 blocks over 4 waves: the last two are requested twice (same bytes, same place)
  constexpr int NBUF = GL ? ring_slots(kBMT) : 2;   // ring slots of TILES x KS KiB
  extern __shared__ __attribute__((aligned(16))) char gemm_smem[];
  u32x4 (*lds)[TILES][KS][64] = reinterpret_cast<u32x4 (*)[TILES][KS][64]>(gemm_smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // K split (gridDim.y slabs): slab kz multiplies chunks [c_lo, c_hi) of K
  const int nchunks_all = (a.K + KC - 1) / KC;
  const int c_lo = (int)(((long)blockIdx.y * nchunks_all) / gridDim.y), c_hi = (int)(((long)(blockIdx.y + 1) * nchunks_all) / gridDim.y);
  const int nchunks = c_hi - c_lo;
  const int mtiles = (a.M + 15) / 16, ntiles = (a.N + 15) / 16;

  // workgroup id -> (row-block pm, column-block pn): XCD x gets a contiguous id range (ids are dealt
  // round-robin to the 8 XCDs), inside which row-blocks vary fastest within groups of kGroupM
  int pid = blockIdx.x;
  const int total = gridDim.x;
  if (total % 8 == 0) pid = (pid & 7) * (total >> 3) + (pid >> 3);
  const int per_group = group_m * nblocks;
  const int group = pid / per_group, first_m = group * group_m;
  const int gsz = min(mblocks - first_m, group_m);
  const int pm = first_m + (pid % per_group) % gsz;
  const int pn = (pid % per_group) / gsz;

  // the stage loader: piece j of thread t is lane (t & 63) of fragment block j*4 + wave
  const u32x4* src[LOADS];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) {
    const int blk = min(j * NWV + wave, TILES * KS - 1), tile = blk / KS, kc = blk % KS;
    const u32x4* p;
    if (tile < kBMT) {
      const int mt = min(pm * kBMT + tile, mtiles - 1);
      p = apack + ((size_t)mt * nchunks_all + c_lo) * 64;
    } else {
      const int idx = tile - kBMT, mat = idx / BNT;
      const int nt = min(pn * BNT + idx % BNT, ntiles - 1);
      p = reinterpret_cast<const u32x4*>(mat ? a.w1 : a.w0) + ((size_t)nt * nchunks_all + c_lo) * 64;
    }
    src[j] = p + (size_t)kc * 64 + lane;
  }
  const int nstages = (nchunks + KS - 1) / KS;
  // Register-staged ring (!GL) — NS staging sets: stages s+1 .. s+NS are in flight while stage s is multiplied.  One set exposed a
  // full memory round trip per stage (1.1 us/stage, profiles/r1_notes.md); two hide it.  Four / six sets on the small tiles were
  // measured in round 3 and lost 10-30 % (profiles/r3_notes.md §7): those tiles were not waiting for memory.
  constexpr int NS = 2;
  constexpr int NU = (NS % 2) ? 2 * NS : NS;     // steps per unrolled round: ring half and set both compile-time
  u32x4 stg[NS][LOADS];
  auto fetch = [&](u32x4 (&stg)[LOADS], int s) {
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
      const int kc = min(j * NWV + wave, TILES * KS - 1) % KS;
      const int c = min(s * KS + kc, nchunks - 1);          // clamped: a chunk past K is loaded but never used
      stg[j] = src[j][(size_t)(c - kc) * 64];
    }
  };
  auto commit = [&](int buf, const u32x4 (&stg)[LOADS]) {
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
      const int blk = min(j * NWV + wave, TILES * KS - 1);
      lds[buf][blk / KS][blk % KS][lane] = stg[j];
    }
  };

  f32x4 chain[NT][kWM][WN], tot[NT][kWM][WN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mi = 0; mi < kWM; ++mi)
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) {
        chain[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        tot[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  auto retire = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mi = 0; mi < kWM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[t][mi][ni][r] = __fadd_rn(tot[t][mi][ni][r], chain[t][mi][ni][r]);
          chain[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
  };
  int seg = 1;                                   // next range boundary: chunk (seg * nchunks) / nw
  int boundary = (seg * nchunks) / nw;

  // A chunk boundary of the decode kernel's K split: retire the running chain (`while`: empty ranges — nw > nchunks — retire
  // zeros, as the decode kernel adds them)
  auto retire_at = [&](int c) {
    while (seg < nw && c == boundary) {
      retire();
      ++seg;
      boundary = (seg * nchunks) / nw;
    }
  };

  if constexpr (GL) {
    // ---- LDS-DMA ring: one chunk per slot, four slots; the fragments of chunk c + 1 are READ while chunk c is multiplied ----
    // Per chunk:
    //   wait: my pieces of chunk c + 1 have landed (chunks c + 2, c + 3 stay in flight) and my fragment reads of chunk c are
    //         complete (they were issued a whole chunk of MFMAs ago) | barrier: everyone's are |
    //   retire check | read chunk c + 1 -> the other fragment set | request chunk c + 4 into chunk c's slot (its fragments are
    //   in registers on every wave: the barrier said so), the requests spread among the MFMAs of chunk c (a request costs
    //   ~60-85 issue cycles, four MFMAs 64).
    // Requests past the last chunk are clamped onto it (into a slot nobody reads again), so the count in front of every barrier
    // is the same.  The waits are hand-counted: an LDS-DMA is invisible to the compiler's s_waitcnt bookkeeping for the
    // ds_reads that follow, and a __syncthreads() would drain vmcnt to 0.
    // How this came about (cycle stamps per phase, profiles/r3_notes.md §7): with register staging a wave of the 128-row tile
    // spent ~750 cycles per two-chunk stage issuing ds_write_b128 (13 LDS-issue cycles per KiB), ~1000 issuing its 8 global
    // loads behind everyone else's and ~1500 in read -> wait -> MFMA chains, for 512 cycles of MFMA issue.
    constexpr int NF = kWM + NT * WN;
    u32x4 fr[2][NF];
    auto dma = [&](int c_req, int slot) {
      const int c = min(c_req, nchunks - 1);
#pragma unroll
      for (int j = 0; j < LOADS; ++j) {
        const int blk = min(j * NWV + wave, TILES - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)c * 64),
                                         (__attribute__((address_space(3))) void*)&lds[slot][blk][0][0], 16, 0, 0);
      }
    };
    auto read = [&](u32x4 (&f)[NF], int slot) {
#pragma unroll
      for (int mi = 0; mi < kWM; ++mi) f[mi] = lds[slot][wm * kWM + mi][0][lane];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) f[kWM + t * WN + ni] = lds[slot][kBMT + t * BNT + wn * WN + ni][0][lane];
    };
    int slot_c = 0;                               // c % NBUF, kept as a counter (NBUF need not be a power of two)
    auto step = [&](u32x4 (&cur)[NF], u32x4 (&nxt)[NF], int c) {
      // `cur` rides through the statement as in/out operands: the compiler then places its own wait for those reads HERE (they
      // were issued a chunk ago) instead of a conservative lgkmcnt(0) behind the next chunk's reads, in front of the first MFMA
      static_assert(NF == 8 || NF == 6 || NF == 5, "operand lists below");
      if constexpr (NF == 8) {
        asm volatile("s_waitcnt vmcnt(%8) lgkmcnt(0)\n\t" UA2_GEMM_BAR
                     : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7])
                     : "n"((UA2_GEMM_DBG & 1) ? 0 : (NBUF - 2) * LOADS)
                     : "memory");
      } else if constexpr (NF == 6) {
        asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)\n\t" UA2_GEMM_BAR
                     : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5])
                     : "n"((UA2_GEMM_DBG & 1) ? 0 : (NBUF - 2) * LOADS)
                     : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%5) lgkmcnt(0)\n\t" UA2_GEMM_BAR
                     : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4])
                     : "n"((UA2_GEMM_DBG & 1) ? 0 : (NBUF - 2) * LOADS)
                     : "memory");
      }
      retire_at(c);                              // in front of the reads: everything behind it is one block
      const int slot_n = slot_c + 1 == NBUF ? 0 : slot_c + 1;
      if constexpr (!(UA2_GEMM_DBG & 4)) read(nxt, slot_n);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(UA2_GEMM_DBG & 1)) dma(c + NBUF, slot_c);
      slot_c = slot_n;
      if constexpr (!(UA2_GEMM_DBG & 2)) {
#pragma unroll
      for (int mi = 0; mi < kWM; ++mi) {
        AFrag<DT> af;
        af.v = __builtin_bit_cast(decltype(af.v), cur[mi]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) af.mma(cur[kWM + t * WN + ni], chain[t][mi][ni]);
      }
      }
      constexpr int MF = kWM * NT * WN;          // MFMAs of the chunk
#pragma unroll
      for (int g = 0; g < LOADS; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, MF / LOADS > 0 ? MF / LOADS : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    };
#pragma unroll
    for (int t = 0; t < NBUF; ++t) dma(t, t);
    static_assert((NBUF - 1) * LOADS <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 1) * LOADS) : "memory");
    read(fr[0], 0);
    for (int c = 0; c < ((UA2_GEMM_DBG & 32) ? 0 : nchunks); c += 2) {
      step(fr[0], fr[1], c);
      if (c + 1 < nchunks) step(fr[1], fr[0], c + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail requests: nothing may land in the ring once the epilogue reuses it
  } else {
    // ---- register staging (the hand-over instantiations and the UA2_GEMM_NO_GLDS hook): two-chunk stages, two ring halves ----
    auto compute = [&](int buf, int s) {
#pragma unroll
      for (int kc = 0; kc < KS; ++kc) {
        const int c = s * KS + kc;
        u32x4 fa[kWM], fb[NT][WN];
#pragma unroll
        for (int mi = 0; mi < kWM; ++mi) fa[mi] = lds[buf][wm * kWM + mi][kc][lane];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) fb[t][ni] = lds[buf][kBMT + t * BNT + wn * WN + ni][kc][lane];
        if constexpr (kBMT == 8) __builtin_amdgcn_sched_barrier(0);   // a chunk's reads before its MFMAs (-2 ... -5 % on the 128-row tile)
        if (c < nchunks) {
          retire_at(c);
#pragma unroll
          for (int mi = 0; mi < kWM; ++mi) {
            AFrag<DT> af;
            af.v = __builtin_bit_cast(decltype(af.v), fa[mi]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int ni = 0; ni < WN; ++ni) af.mma(fb[t][ni], chain[t][mi][ni]);
          }
        }
      }
    };
    // stage t travels in set t % NS: stage 0 goes straight to the ring, stages 1 .. NS are requested behind it
    fetch(stg[0], 0);
    commit(0, stg[0]);
#pragma unroll
    for (int t = 1; t <= NS; ++t)
      if (t < nstages) fetch(stg[t % NS], t);
    // ua2_lds_barrier, not __syncthreads(): in front of an s_barrier it can see, the compiler drains vmcnt to 0 — with it
    // the register staging sets never had more than one stage of MFMAs to cover a memory round trip
    ua2_lds_barrier();
    for (int s0 = 0; s0 < nstages; s0 += NU) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int s = s0 + u;
        if (s >= nstages) break;
        // the other ring half was read in stage s - 1 and everyone has passed the barrier since: fill it first, so that the
        // ds_write latency runs under this stage's MFMAs instead of in front of the barrier
        if (s + 1 < nstages) commit((u + 1) & 1, stg[(u + 1) % NS]);
        if (s + 1 + NS < nstages) fetch(stg[(u + 1) % NS], s + 1 + NS);
        compute(u & 1, s);
        ua2_lds_barrier();
      }
    }
  }
  while (seg <= nw) { retire(); ++seg; }         // the last range (and any empty ones after it)
  if constexpr ((UA2_GEMM_DBG & 16) != 0) {
    if (tot[0][0][0][0] == 1.2345f) a.y[0] = 0.f;   // keeps the loop alive
    return;
  }

  // ---- QKV epilogue, staged (half-split RoPE, head_size 128: the workgroup's 128 columns are exactly one head) ----
  // The generic epilogue below issues, per output element, two scalar loads of the RoPE table and one 4-byte (q) or
  // 2-byte (K/V) store into 32- / 16-byte runs: at 6240 prefill rows that was 2/3 of the launch (838 us against 280 us of
  // MFMA work, profiles/r2_notes.md).  Here every wave parks its un-rotated 64 x 64 patch in the (now idle) operand ring,
  // in natural dim order, and walks it row by row: a lane owns 4 consecutive dims of one row, reads its rotation partner
  // from LDS, the cos / sin of its 4 dims with one 16-byte load each, and stores 16 bytes (q, fp32) or 8 bytes (K/V,
  // bf16) — 128- / 64-byte runs.  Same operations in the same order as linear_epilogue: identical bits.
  if constexpr (EPI == UA2_EPI_QKV_ROPE) {
    if (a.rope_mode == UA2_ROPE_HALF_SPLIT && a.kv.head_size == 128 && !a.bias) {
      __syncthreads();                                   // every wave is done with the operand ring
      float* patch = reinterpret_cast<float*>(&lds[0][0][0][0]) + (size_t)wave * (kWM * 16) * 64;
      const int colq = lane & 15, gq = lane >> 4;
#pragma unroll
      for (int mi = 0; mi < kWM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int prow = mi * 16 + 4 * gq + r;
            const int pcol = (colq < 8) ? ni * 8 + colq : 32 + ni * 8 + (colq - 8);   // [0,32): dims 32 wn + ..; [32,64): 64 + 32 wn + ..
            patch[prow * 64 + pcol] = tot[0][mi][ni][r];
          }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int h = pn;                                  // head index of this workgroup's column block
      const int hs = 128, half = 64;
      const bool is_q = h < a.kv.n_head, is_k = !is_q && h < a.kv.n_head + a.kv.n_kv;
      const bool rot = is_q || is_k;
      const int kvh = is_q ? 0 : (is_k ? h - a.kv.n_head : h - a.kv.n_head - a.kv.n_kv);
      const int j = lane & 15, jj = j & 7;
      const bool hi = j >= 8;
      const int d0 = 32 * wn + 4 * jj;                   // table column / low-half dim of this lane's 4 dims
      for (int it = 0; it < kWM * 4; ++it) {
        const int prow = it * 4 + gq;
        const int m = (pm * kBMT + wm * kWM) * 16 + prow;
        if (m >= a.M) continue;                          // uniform over the row's 16 lanes
        float4 own = *reinterpret_cast<const float4*>(patch + prow * 64 + (hi ? 32 : 0) + 4 * jj);
        float4 oth = *reinterpret_cast<const float4*>(patch + prow * 64 + (hi ? 0 : 32) + 4 * jj);
        if (a.prologue == UA2_PRO_SCALED) {              // the row scale first, as linear_epilogue does
          const float rs = scaled_rstd_row(a, m, j);      // the row's 16 lanes together (the ring holds the patches: no LDS to spare)
          own.x = __fmul_rn(own.x, rs); own.y = __fmul_rn(own.y, rs); own.z = __fmul_rn(own.z, rs); own.w = __fmul_rn(own.w, rs);
          oth.x = __fmul_rn(oth.x, rs); oth.y = __fmul_rn(oth.y, rs); oth.z = __fmul_rn(oth.z, rs); oth.w = __fmul_rn(oth.w, rs);
        }
        const int pos = a.row_pos[m];
        float4 out = own;
        if (rot) {
          const float4 cs = *reinterpret_cast<const float4*>(a.rope_cos + (size_t)pos * half + d0);
          const float4 sn = *reinterpret_cast<const float4*>(a.rope_sin + (size_t)pos * half + d0);
          // lo half: x1 cos + (-x2) sin ; hi half: x2 cos + x1 sin  (lit_model.py:795-806), products rounded separately
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
          void* pool = is_k ? a.kv.k_pool : a.kv.v_pool;
          if constexpr (DT == UA2_BF16) {
            uint2 pk;
            pk.x = (unsigned)f2bf(out.x) | ((unsigned)f2bf(out.y) << 16);
            pk.y = (unsigned)f2bf(out.z) | ((unsigned)f2bf(out.w) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(pool) + base) = pk;
          } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(pool) + base) = out;
          }
        }
      }
      return;
    }
    // No rotation, head_size 64 (the codec's DiT: fused q|k|v with bias): the same staging — a wave's 64 columns are exactly one
    // head, a lane owns 4 consecutive dims of a row — without the rotation.  The generic path below took 52 us of a launch
    // whose GEMM is worth 33 (profiles/r2_notes.md §4).  Same operations (sum + bias, one rounding): identical bits.
    if (a.rope_mode == UA2_ROPE_NONE && a.kv.head_size == 64) {
      __syncthreads();
      float* patch = reinterpret_cast<float*>(&lds[0][0][0][0]) + (size_t)wave * (kWM * 16) * 64;
      const int colq = lane & 15, gq = lane >> 4;
#pragma unroll
      for (int mi = 0; mi < kWM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) patch[(mi * 16 + 4 * gq + r) * 64 + ni * 16 + colq] = tot[0][mi][ni][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int hs = 64;
      const int h = pn * 2 + wn;                         // head of this wave's 64 columns
      if (h * hs >= a.N) return;                         // wave-uniform: past the last head
      const bool is_q = h < a.kv.n_head, is_k = !is_q && h < a.kv.n_head + a.kv.n_kv;
      const int kvh = is_q ? 0 : (is_k ? h - a.kv.n_head : h - a.kv.n_head - a.kv.n_kv);
      const int d0 = 4 * colq;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + (size_t)h * hs + d0);
      for (int it = 0; it < kWM * 4; ++it) {
        const int prow = it * 4 + gq;
        const int m = (pm * kBMT + wm * kWM) * 16 + prow;
        if (m >= a.M) continue;
        float4 out = *reinterpret_cast<const float4*>(patch + prow * 64 + d0);
        if (a.bias) { out.x = __fadd_rn(out.x, b4.x); out.y = __fadd_rn(out.y, b4.y); out.z = __fadd_rn(out.z, b4.z); out.w = __fadd_rn(out.w, b4.w); }
        if (is_q) {
          *reinterpret_cast<float4*>(a.q_out + (size_t)m * a.kv.n_head * hs + (size_t)h * hs + d0) = out;
        } else {
          const int pos = a.row_pos[m];
          const int page = a.kv.page_table[(size_t)kv_table_row(a, m) * a.kv.max_pages + ua2_page_slot(a.kv, pos)];
          const size_t base = (((size_t)page * a.kv.n_kv + kvh) * UA2_PAGE + (pos % UA2_PAGE)) * hs + d0;
          void* pool = is_k ? a.kv.k_pool : a.kv.v_pool;
          if constexpr (DT == UA2_BF16) {
            uint2 pk;
            pk.x = (unsigned)f2bf(out.x) | ((unsigned)f2bf(out.y) << 16);
            pk.y = (unsigned)f2bf(out.z) | ((unsigned)f2bf(out.w) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(pool) + base) = pk;
          } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(pool) + base) = out;
          }
        }
      }
      return;
    }
    // Other RoPE flavours / head sizes (Moshi family): the generic per-element epilogue, but fed from LDS.  Fed from the
    // accumulator registers, its control flow (shuffles, early exits per row) made the compiler keep `tot` in scratch memory
    // for the WHOLE kernel — 605 scratch instructions, every retire() a round trip: the round-1 QKV launch ran at a third of
    // the SwiGLU launch's rate for that reason alone (profiles/r2_notes.md).
    __syncthreads();
    float* patch = reinterpret_cast<float*>(&lds[0][0][0][0]) + (size_t)wave * (kWM * 16) * 64;
#pragma unroll
    for (int mi = 0; mi < kWM; ++mi)
#pragma unroll
      for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[((mi * WN + ni) * 4 + r) * 64 + lane] = tot[0][mi][ni][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int colg = lane & 15, gg = lane >> 4;
#pragma unroll 1
    for (int mi = 0; mi < kWM; ++mi) {
      const int m0 = (pm * kBMT + wm * kWM + mi) * 16;
      if (m0 >= a.M) continue;                    // wave-uniform
      const int rows = min(16, a.M - m0);
#pragma unroll 1
      for (int ni = 0; ni < WN; ++ni) {
        const int nt = pn * BNT + wn * WN + ni;
        if (nt >= ntiles) continue;               // wave-uniform
        int tile[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) tile[t] = nt;
        EpiPre pre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          epilogue_prefetch<DT, EPI>(a, nt, 4 * gg + r, colg, pre[r], m0);
          if (a.prologue == UA2_PRO_SCALED) pre[r].rstd = scaled_rstd_row(a, m0 + 4 * gg + r, colg);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v[NT];
          v[0] = patch[((mi * WN + ni) * 4 + r) * 64 + lane];
          linear_epilogue<DT, EPI, NT, HO>(a, v, tile, 4 * gg + r, colg, pre[r], m0, rows);
        }
      }
    }
    return;
  }

  // ---- staged epilogue (STORE / RESIDUAL / SWIGLU / GELU on whole wave patches) ----
  // Fed from the accumulator registers, the per-element epilogue below stores 4 bytes per lane into 64-byte runs, loads the
  // residual the same way and re-reads bias / out_scale / the following norm's weight once per element: 17-31 % of a launch at
  // 6272 rows and 30-45 % at the DiT's 1000 (tools/ubench/gemm_knock2.sh, profiles/r4_notes.md §9).  Here every wave parks its
  // patch (kWM x 16 rows x 64 columns, fp32) in the idle operand ring in natural order and walks it by rows: a lane owns 4
  // consecutive columns, so results, residual, bias and scales move as 16-byte pieces in 256-byte runs (the packed operand
  // hand-off as 8-byte pieces).  Every value goes through the same operations in the same order as linear_epilogue: same bits
  // (tests/test_gpu_invariance.py compares the two forms and the decode kernel).  Launches with partial arg-max outputs or with
  // N not a multiple of the wave's column span take the per-element form.
  if constexpr (EPI != UA2_EPI_QKV_ROPE) {
    constexpr int SPAN = WN * 16;                        // columns of one matrix a wave owns: 64 (32 for SWIGLU)
    const bool slab = (flags & 2) != 0;                  // K split (ua2hip.h split_ws): this workgroup's partial sums, raw, into its slab
    if ((flags & 1) && a.N % SPAN == 0 && !a.part_max) {
      __syncthreads();                                   // every wave is done with the operand ring
      float* patch = reinterpret_cast<float*>(gemm_smem) + (size_t)wave * (kWM * 16) * 64;
      {
        const int colq = lane & 15, gq = lane >> 4;
#pragma unroll
        for (int mi = 0; mi < kWM; ++mi)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
              for (int r = 0; r < 4; ++r) patch[(mi * 16 + 4 * gq + r) * 64 + (t * WN + ni) * 16 + colq] = tot[t][mi][ni][r];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      constexpr int LPR = SPAN / 4;                      // lanes per row: 16 (8 for SWIGLU: a 16-lane group walks two rows at once)
      constexpr int RPI = 64 / LPR;                      // rows per iteration: 4 (8)
      constexpr int ITERS = kWM * 16 / RPI;
      const int j = lane & (LPR - 1), rsub = lane / LPR;
      const int n0 = (pn * BNT + wn * WN) * 16 + 4 * j;  // this lane's 4 columns (of each matrix)
      if (n0 >= a.N) return;                             // wave-uniform (N % SPAN == 0)
      const int mbase = (pm * kBMT + wm * kWM) * 16;
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 b0 = zero4, b1 = zero4, os4 = make_float4(1.f, 1.f, 1.f, 1.f), nw4 = zero4;
      if (a.bias) b0 = *reinterpret_cast<const float4*>(a.bias + n0);
      if constexpr (NT == 2) { if (a.bias && a.bias1) b1 = *reinterpret_cast<const float4*>(a.bias1 + n0); }
      if constexpr (EPI == UA2_EPI_RESIDUAL) { if (a.out_scale) os4 = *reinterpret_cast<const float4*>(a.out_scale + n0); }
      if constexpr (HO) { if (a.y_norm_w) nw4 = *reinterpret_cast<const float4*>(a.y_norm_w + n0); }
      float4 res[ITERS];
      if constexpr (EPI == UA2_EPI_RESIDUAL) {           // every residual piece of the patch requested before the first store
        if (!slab) {
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
        float rs = 1.f;
        if (a.prologue == UA2_PRO_SCALED) {              // the row's scale, by the 16 lanes of a group together
          if constexpr (NT == 1) {
            rs = scaled_rstd_row(a, m, lane & 15);
          } else {                                       // the group's two rows one after the other; each half keeps its own
            const int mg = mbase + it * RPI + 2 * (lane >> 4);
            const float ra = scaled_rstd_row(a, mg, lane & 15), rb = scaled_rstd_row(a, mg + 1, lane & 15);
            rs = (rsub & 1) ? rb : ra;
          }
        }
        if (m >= a.M) continue;
        float4 v0 = *reinterpret_cast<const float4*>(patch + prow * 64 + 4 * j);
        if constexpr (EPI == UA2_EPI_RESIDUAL) {
          if (slab) {
            *reinterpret_cast<float4*>(a.split_ws + ((size_t)blockIdx.y * a.M + m) * a.N + n0) = v0;
            continue;
          }
        }
        float4 v1 = zero4;
        if constexpr (NT == 2) v1 = *reinterpret_cast<const float4*>(patch + prow * 64 + SPAN + 4 * j);
        if (a.prologue == UA2_PRO_SCALED) {
          v0.x = __fmul_rn(v0.x, rs); v0.y = __fmul_rn(v0.y, rs); v0.z = __fmul_rn(v0.z, rs); v0.w = __fmul_rn(v0.w, rs);
          if constexpr (NT == 2) { v1.x = __fmul_rn(v1.x, rs); v1.y = __fmul_rn(v1.y, rs); v1.z = __fmul_rn(v1.z, rs); v1.w = __fmul_rn(v1.w, rs); }
        }
        if (a.bias) {
          v0.x = __fadd_rn(v0.x, b0.x); v0.y = __fadd_rn(v0.y, b0.y); v0.z = __fadd_rn(v0.z, b0.z); v0.w = __fadd_rn(v0.w, b0.w);
          if constexpr (NT == 2) { v1.x = __fadd_rn(v1.x, b1.x); v1.y = __fadd_rn(v1.y, b1.y); v1.z = __fadd_rn(v1.z, b1.z); v1.w = __fadd_rn(v1.w, b1.w); }
        }
        float4 out = v0;
        if constexpr (EPI == UA2_EPI_RESIDUAL) {
          if (a.out_scale) { out.x = __fmul_rn(os4.x, v0.x); out.y = __fmul_rn(os4.y, v0.y); out.z = __fmul_rn(os4.z, v0.z); out.w = __fmul_rn(os4.w, v0.w); }
          out.x = __fadd_rn(out.x, res[it].x); out.y = __fadd_rn(out.y, res[it].y); out.z = __fadd_rn(out.z, res[it].z); out.w = __fadd_rn(out.w, res[it].w);
        } else if constexpr (EPI == UA2_EPI_SWIGLU) {
          out.x = ua2_act_glu(a, v0.x, v1.x); out.y = ua2_act_glu(a, v0.y, v1.y); out.z = ua2_act_glu(a, v0.z, v1.z); out.w = ua2_act_glu(a, v0.w, v1.w);
        } else if constexpr (EPI == UA2_EPI_GELU) {
          out.x = ua2_act_gelu(a, v0.x); out.y = ua2_act_gelu(a, v0.y); out.z = ua2_act_gelu(a, v0.z); out.w = ua2_act_gelu(a, v0.w);
        }
        if (a.y) *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n0) = out;
        if constexpr (EPI == UA2_EPI_SWIGLU || EPI == UA2_EPI_GELU) {
          if (a.y_packed) store_packed4<DT>(a.y_packed, m, n0, a.N / KC, out);
        }
        if constexpr (HO && (EPI == UA2_EPI_STORE || EPI == UA2_EPI_RESIDUAL)) {
          if (a.y_norm_w) {
            // ssq_tile16's tree over the tile's 16 columns (xor 1, 2, 4, 8) with 4 columns per lane: two levels in registers, two across lanes
            float s = __fadd_rn(__fmaf_rn(out.x, out.x, __fmul_rn(out.y, out.y)), __fmaf_rn(out.z, out.z, __fmul_rn(out.w, out.w)));   // first level fused, as ssq_tile16 spells it
            s = __fadd_rn(s, __shfl_xor(s, 1));
            s = __fadd_rn(s, __shfl_xor(s, 2));
            if (a.y_ssq && (j & 3) == 0) a.y_ssq[(size_t)m * ((a.N + 15) >> 4) + (n0 >> 4)] = s;
            const float4 h = make_float4(__fmul_rn(out.x, nw4.x), __fmul_rn(out.y, nw4.y), __fmul_rn(out.z, nw4.z), __fmul_rn(out.w, nw4.w));
            if (a.y_h) store_row4<DT>(a.y_h, (size_t)m * a.ldh + n0, h);
            if (a.y_packed) store_packed4<DT>(a.y_packed, m, n0, a.N / KC, h);
          }
        }
      }
      return;
    }
  }

  // ---- epilogue: lane holds D[row = 4*(lane >> 4) + r][col = lane & 15] of each 16 x 16 tile ----
  // The epilogue's loads (residual; position -> RoPE table entry / page id) are gathered for the four rows of a
  // tile before any of its stores, so the four dependent chains overlap instead of running one after the other
  // (a store between two loads pins their order: 250 us of the 380 us QKV launch at M = 2048 before this).
  const int col = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < kWM; ++mi) {
    const int m0 = (pm * kBMT + wm * kWM + mi) * 16;
    if (m0 >= a.M) continue;                      // wave-uniform
    const int rows = min(16, a.M - m0);
    EpiPre row_pre[4];
    if constexpr (EPI == UA2_EPI_QKV_ROPE || EPI == UA2_EPI_STORE) {   // per-row part (position, table row, forbid): once per row tile
#pragma unroll
      for (int r = 0; r < 4; ++r) epilogue_prefetch_a<DT, EPI>(a, 0, 4 * g + r, col, row_pre[r], m0);
    }
    float row_rstd[4] = {1.f, 1.f, 1.f, 1.f};                          // UA2_PRO_SCALED: once per row tile, the row's 16 lanes together
    if (a.prologue == UA2_PRO_SCALED) {
#pragma unroll
      for (int r = 0; r < 4; ++r) row_rstd[r] = scaled_rstd_row(a, m0 + 4 * g + r, col);
    }
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
      const int nt = pn * BNT + wn * WN + ni;
      if (nt >= ntiles) continue;                 // wave-uniform
      int tile[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) tile[t] = nt;
      EpiPre pre[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pre[r] = row_pre[r];
        if constexpr (EPI != UA2_EPI_QKV_ROPE && EPI != UA2_EPI_STORE) epilogue_prefetch_a<DT, EPI>(a, nt, 4 * g + r, col, pre[r], m0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        epilogue_prefetch_b<DT, EPI>(a, nt, 4 * g + r, col, pre[r], m0);
        pre[r].rstd = row_rstd[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = tot[t][mi][ni][r];
        linear_epilogue<DT, EPI, NT, HO>(a, v, tile, 4 * g + r, col, pre[r], m0, rows);
      }
    }
  }
}


// ---- skinny form (a few dozen rows: batched decode) ---------------------------------------------------
// The decode kernel's own structure with the LDS row tile replaced by the packed operand in L2: workgroup =
// one 16-column weight tile x up to 64 rows, K split over `nw` waves exactly as ua2_gemv.hip splits it, every
// weight fragment loaded once (non-temporal, straight from HBM) and used for MT row tiles; partial sums meet
// in LDS and are added in wave order.  HBM-bound on the weights like the decode kernel; the operand rows
// (M x K, a few hundred KiB) are re-read by every workgroup from L2.
constexpr int kSkinnyMT = 4;

template <int DT, int EPI, int CPW>
__global__ __launch_bounds__(1024) void skinny_kernel(const ua2_linear_args a, const u32x4* __restrict__ apack) {
  constexpr int KC = Elem<DT>::KC;
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr int MT = kSkinnyMT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);               // [nw][NT][MT][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  float* rstd_l = red + (size_t)nw * NT * MT * 256;          // [MT * 16] row scales (UA2_PRO_SCALED)
  const int nchunks = (a.K + KC - 1) / KC;
  const int mtiles = (a.M + 15) / 16;
  const int nt = blockIdx.x, mt0 = blockIdx.y * MT;
  if (a.prologue == UA2_PRO_SCALED) scaled_rstd_rows(a, mt0 * 16, min(MT * 16, a.M - mt0 * 16), rstd_l, tid, blockDim.x);
  const u32x4* wp[NT];
  wp[0] = reinterpret_cast<const u32x4*>(a.w0) + (size_t)nt * nchunks * 64 + lane;
  if constexpr (NT == 2) wp[1] = reinterpret_cast<const u32x4*>(a.w1) + (size_t)nt * nchunks * 64 + lane;
  const u32x4* ap[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) ap[mi] = apack + (size_t)min(mt0 + mi, mtiles - 1) * nchunks * 64 + lane;
  const int c0 = (wave * nchunks) / nw, c1 = ((wave + 1) * nchunks) / nw;
  const int last = max(c1 - 1, 0);

  f32x4 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[t][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int cb = c0; cb < c1; cb += CPW) {
    u32x4 wf[NT][CPW], af[MT][CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      const size_t off = (size_t)min(cb + u, last) * 64;
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[t][u] = __builtin_nontemporal_load(wp[t] + off);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) af[mi][u] = ap[mi][off];
    }
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      if (cb + u < c1) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          AFrag<DT> f;
          f.v = __builtin_bit_cast(decltype(f.v), af[mi][u]);
#pragma unroll
          for (int t = 0; t < NT; ++t) f.mma(wf[t][u], acc[t][mi]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) *reinterpret_cast<f32x4*>(&red[(((wave * NT + t) * MT) + mi) * 256 + lane * 4]) = acc[t][mi];
  __syncthreads();
  if (tid >= 256) return;
  const int row = tid >> 4, col = tid & 15;
  const int srcl = (((row >> 2) << 4) + col) * 4 + (row & 3);
  int tile[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tile[t] = nt;
  EpiPre pre[MT];                                            // all row tiles' epilogue loads before any store
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) epilogue_prefetch_a<DT, EPI>(a, nt, row, col, pre[mi], (mt0 + mi) * 16);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) epilogue_prefetch_b<DT, EPI>(a, nt, row, col, pre[mi], (mt0 + mi) * 16);
  if (a.prologue == UA2_PRO_SCALED) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) pre[mi].rstd = rstd_l[mi * 16 + row];
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int m0 = (mt0 + mi) * 16;
    if (m0 >= a.M) break;                                    // uniform over the workgroup
    float v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float sacc = 0.f;
      for (int w = 0; w < nw; ++w) sacc += red[(((w * NT + t) * MT) + mi) * 256 + srcl];
      v[t] = sacc;
    }
    linear_epilogue<DT, EPI, NT>(a, v, tile, row, col, pre[mi], m0, min(16, a.M - m0));
  }
}

template <int DT, int EPI>
void launch_skinny(const ua2_linear_args& a, ua2_gemv_geometry geo, hipStream_t s) {
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr auto kern = skinny_kernel<DT, EPI, 4>;
  ua2_allow_big_lds<kern>();
  const size_t smem = (size_t)geo.waves * NT * kSkinnyMT * 256 * sizeof(float) + kSkinnyMT * 16 * sizeof(float);
  const dim3 grid(ua2_ceil_div(a.N, 16), ua2_ceil_div(ua2_ceil_div(a.M, 16), kSkinnyMT));
  hipLaunchKernelGGL(kern, grid, dim3(geo.waves * 64), smem, s, a, reinterpret_cast<const u32x4*>(a.x_packed ? a.x_packed : a.workspace));
}

// ---- K split (ua2hip.h split_ws): y = resid + out_scale (.) ((((s0 + s1) + s2) + s3) + bias), the slabs in index order ----
// Same operations as linear_epilogue's RESIDUAL branch on the combined sum.
__global__ __launch_bounds__(256) void splitk_combine_kernel(const ua2_linear_args a, const int slabs) {
  const int n4 = a.N >> 2;
  const size_t total = (size_t)a.M * n4, slab_elems = (size_t)a.M * a.N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
    const float* p = a.split_ws + (size_t)m * a.N + n;
    float4 t = *reinterpret_cast<const float4*>(p);
    for (int k = 1; k < slabs; ++k) {
      const float4 u = *reinterpret_cast<const float4*>(p + (size_t)k * slab_elems);
      t.x = __fadd_rn(t.x, u.x); t.y = __fadd_rn(t.y, u.y); t.z = __fadd_rn(t.z, u.z); t.w = __fadd_rn(t.w, u.w);
    }
    if (a.bias) {
      const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
      t.x = __fadd_rn(t.x, b.x); t.y = __fadd_rn(t.y, b.y); t.z = __fadd_rn(t.z, b.z); t.w = __fadd_rn(t.w, b.w);
    }
    if (a.out_scale) {
      const float4 g = *reinterpret_cast<const float4*>(a.out_scale + n);
      t.x = __fmul_rn(g.x, t.x); t.y = __fmul_rn(g.y, t.y); t.z = __fmul_rn(g.z, t.z); t.w = __fmul_rn(g.w, t.w);
    }
    const float4 r = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n);
    t.x = __fadd_rn(t.x, r.x); t.y = __fadd_rn(t.y, r.y); t.z = __fadd_rn(t.z, r.z); t.w = __fadd_rn(t.w, r.w);
    *reinterpret_cast<float4*>(a.y + (size_t)m * a.ldy + n) = t;
  }
}

template <int DT, int PRO, int NVW, int MAXV, int ROWS>
void launch_prep16(const ua2_linear_args& a, hipStream_t s) {
  constexpr auto kern = gemm_prep16_kernel<DT, PRO, NVW, MAXV, ROWS>;
  const int kc = Elem<DT>::KC, nchunks = ua2_ceil_div(a.K, kc);
  size_t smem = 0;
  if constexpr (ROWS == 16) {
    ua2_allow_big_lds<kern>();
    smem = (size_t)16 * ((size_t)nchunks * kc * Elem<DT>::BYTES + 16);
  }
  // the grid covers whole row tiles: rows >= M of the last tile are zero-filled
  hipLaunchKernelGGL(kern, dim3(ua2_ceil_div(a.M, 16) * (16 / ROWS)), dim3(64 * ROWS), smem, s, a, a.workspace);
}

template <int DT, int PRO>
void launch_prep(const ua2_linear_args& a, int nthreads, hipStream_t s) {
  // the wave-per-row forms where a row fits its registers (and, for the 16-row form, the tile its LDS image), for the decode
  // kernel's usual wave counts: 16 rows per workgroup from 128 row tiles up (whole fragment blocks written), 4 rows per workgroup
  // below (the 16-workgroup launch at 256 rows cost B = 256 decode 1 ms/frame with its 66 NORM preps).
  // UA2_GEMM_PREP16_MIN_ROWS / UA2_GEMM_PREP4_MIN_ROWS override the thresholds; UA2_GEMM_OLD_PREP selects the per-row kernel (test hooks: same bits).
  const int nvw = nthreads / 64;
  const char* mr_env = getenv("UA2_GEMM_PREP16_MIN_ROWS");   // read per call (launches are captured into graphs: not a per-frame cost)
  const int min_rows = mr_env ? atoi(mr_env) : 2048;
  // ... and below ~1000 rows the per-row kernel (a workgroup of `waves` waves per row) still wins: at 256 rows the 4-row form's
  // 64 workgroups cost B = 256 decode 0.8 ms/frame (12.2 against 11.5); at the DiT's 1000 rows it saves ~1 us per prep.
  const char* m4_env = getenv("UA2_GEMM_PREP4_MIN_ROWS");
  const int min_rows4 = m4_env ? atoi(m4_env) : 960;
  const bool fits = a.K % 4 == 0 && a.K <= (DT == UA2_BF16 ? 4096 : 2048) && !getenv("UA2_GEMM_OLD_PREP") && (a.M >= min_rows || a.M >= min_rows4);
  if (fits) {
    const bool small = a.K <= 2048, tile16 = a.M >= min_rows;
    auto go = [&](auto nvw_c) {
      constexpr int W = decltype(nvw_c)::value;
      if (tile16) { if (small) launch_prep16<DT, PRO, W, 8, 16>(a, s); else launch_prep16<DT, PRO, W, 16, 16>(a, s); }
      else { if (small) launch_prep16<DT, PRO, W, 8, 4>(a, s); else launch_prep16<DT, PRO, W, 16, 4>(a, s); }
    };
    if constexpr (PRO == UA2_PRO_CAST) { go(std::integral_constant<int, 1>{}); return; }
    switch (nvw) {
      case 4: go(std::integral_constant<int, 4>{}); return;
      case 6: go(std::integral_constant<int, 6>{}); return;
      case 8: go(std::integral_constant<int, 8>{}); return;
      case 12: go(std::integral_constant<int, 12>{}); return;
      case 16: go(std::integral_constant<int, 16>{}); return;
      default: break;
    }
  }
  hipLaunchKernelGGL((gemm_prep_kernel<DT, PRO>), dim3(ua2_ceil_div(a.M, 16) * 16), dim3(nthreads), 0, s, a, a.workspace);
}

template <int DT, int EPI>
void launch_gemm(const ua2_linear_args& a, int nw, hipStream_t s) {
  constexpr int NT = (EPI == UA2_EPI_SWIGLU) ? 2 : 1;
  constexpr int BNT = 2 * (4 / NT);
  const int mtiles = ua2_ceil_div(a.M, 16), nblocks = ua2_ceil_div(ua2_ceil_div(a.N, 16), BNT);
  static const int group_m = getenv("UA2_GEMM_GROUP_M") ? std::max(1, atoi(getenv("UA2_GEMM_GROUP_M"))) : kGroupM;   // experiment hook
  const char* bmt_env = getenv("UA2_GEMM_BMT");                                                                       // experiment / test hook: 4 or 8 (read per call)
  const int force_bmt = bmt_env ? atoi(bmt_env) : 0;

  // 64-row tiles when the 128-row grid cannot give every CU two workgroups (measured on the DiT, M = 1000: 12.0 -> 9.5 ms per
  // step; no change at 2048 rows); 32-row tiles when even those leave CUs idle (M = 1000 x N = 1536: 192 workgroups)
  const int64_t g8 = (int64_t)ua2_ceil_div(mtiles, 8) * nblocks, g4 = (int64_t)ua2_ceil_div(mtiles, 4) * nblocks;
  constexpr bool kCanHo = (EPI == UA2_EPI_STORE || EPI == UA2_EPI_RESIDUAL);
  const bool ho = kCanHo && a.y_norm_w != nullptr;
  const bool no_glds = getenv("UA2_GEMM_NO_GLDS") != nullptr;                                                          // experiment hook: register staging everywhere
  // 64-row tiles already from 192 workgroups (was 256): measured INSIDE the DiT step, where every launch starts on weights
  // that are in no cache — FF2 (1000 x 1536, K = 6144: 192 workgroups of 64 rows against 384 of 32) 57 -> 43 us, the
  // O-projection likewise: 6.50 -> 6.05 ms per step.  (On MALL-warm weights, tools/ubench/gemm_shapes.py, the two tiles are
  // within 3 us of each other — profiles/r4_notes.md §12.)  UA2_GEMM_G8_MIN / UA2_GEMM_G4_MIN: sweep hooks for the two thresholds.
  const char* g8e = getenv("UA2_GEMM_G8_MIN");
  const char* g4e = getenv("UA2_GEMM_G4_MIN");
  const int64_t g8_min = g8e ? atoi(g8e) : 512, g4_min = g4e ? atoi(g4e) : 192;
  const int bmt = force_bmt ? force_bmt : (g8 >= g8_min ? 8 : (g4 >= g4_min ? 4 : 2));
  const u32x4* ap = reinterpret_cast<const u32x4*>(a.x_packed ? a.x_packed : a.workspace);
  auto go = [&](auto bmt_c, auto ho_c, auto gl_c) {
    constexpr int B = decltype(bmt_c)::value;
    constexpr bool H = decltype(ho_c)::value, G = decltype(gl_c)::value;
    constexpr int TILES = B + NT * BNT;
    constexpr auto kern = gemm_kernel<DT, EPI, B, H, G>;
    ua2_allow_big_lds<kern>();
    const int mblocks = ua2_ceil_div(mtiles, B);
    size_t smem = (size_t)(G ? ring_slots(B) : 2 * kKS) * TILES * 1024;
    smem = std::max(smem, (size_t)4 * (B / 2) * 16 * 64 * sizeof(float));   // the staged epilogues park a 64-column patch per wave in the ring
    int ks = 1;
#ifdef UA2_GEMM_EXPERIMENTS   // timing-only hooks with WRONG results (every slab runs the full epilogue on its partial sums): experiment builds only
    if (const char* ks_env = getenv("UA2_GEMM_KSPLIT_HACK")) ks = std::max(1, atoi(ks_env));
#endif
    int split_flags = 0;
    if constexpr (EPI == UA2_EPI_RESIDUAL && !H) {
      // K split proper (ua2hip.h split_ws): long K, a grid that leaves the device short of work, scratch for the slabs.  S depends on
      // (M, N, K) through the grid AND on how many slabs the caller's scratch holds (`fit`): a launch's bits are a function of
      // (M, N, K, split_ws_bytes) — another window length or a smaller scratch is another summation order (documented in ua2hip.h).  Measured inside the DiT step (FF2, 1000 x 1536, K = 6144, 192 workgroups): 6.06 -> 5.49 ms per step
      // with 4 slabs before the combine launch, 5.74 with 2 (timing-only hook UA2_GEMM_KSPLIT_LONGK, profiles/r4_notes.md §12).
      const int64_t grid1 = (int64_t)mblocks * nblocks;
      if (a.split_ws && a.prologue != UA2_PRO_SCALED && !a.part_max && a.N % 64 == 0 && a.ldr % 4 == 0 && a.ldy % 4 == 0 &&
          ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.resid) | reinterpret_cast<uintptr_t>(a.bias) |
            reinterpret_cast<uintptr_t>(a.out_scale) | reinterpret_cast<uintptr_t>(a.split_ws)) & 15) == 0 &&
          ua2_ceil_div(a.K, Elem<DT>::KC) >= (getenv("UA2_GEMM_KSPLIT_MIN_CHUNKS") ? atoi(getenv("UA2_GEMM_KSPLIT_MIN_CHUNKS")) : 128) &&
          grid1 < 512 && !getenv("UA2_GEMM_NO_KSPLIT")) {
        const int want = (int)std::min<int64_t>(4, (768 + grid1 - 1) / grid1);
        const int fit = (int)std::min<size_t>(4, a.split_ws_bytes / ((size_t)a.M * a.N * sizeof(float)));
        if (std::min(want, fit) > 1) { ks = std::min(want, fit); split_flags = 2; }
      }
    }
#ifdef UA2_GEMM_EXPERIMENTS
    if (const char* e2 = getenv("UA2_GEMM_KSPLIT_LONGK"))   // the same, only for RESIDUAL launches with K >= 4096 (the DiT's FF2 inside the step)
      if (EPI == UA2_EPI_RESIDUAL && ua2_ceil_div(a.K, Elem<DT>::KC) >= 128 && !split_flags) ks = std::max(1, atoi(e2));
#endif
    // the staged epilogue moves y / resid / bias / out_scale / y_norm_w as 16-byte pieces: row strides in whole float4s, 16-byte bases
    // (everything torch hands over is; a caller's odd view takes the per-element form: same bits)
    auto al16 = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
    const bool vec_ok = a.ldy % 4 == 0 && al16(a.y) && (EPI != UA2_EPI_RESIDUAL || (a.ldr % 4 == 0 && al16(a.resid) && al16(a.out_scale))) &&
                        al16(a.bias) && al16(a.bias1) && al16(a.y_norm_w) && (!a.y_h || a.ldh % 4 == 0) && al16(a.y_h);
    const int flags = (((getenv("UA2_GEMM_OLD_EPI") && !split_flags) || (!vec_ok && !split_flags)) ? 0 : 1) | split_flags;   // UA2_GEMM_OLD_EPI: test hook, the per-element epilogue everywhere (same bits)
    hipLaunchKernelGGL(kern, dim3(mblocks * nblocks, ks), dim3(256), smem, s, a, ap, ks > 1 ? 1 : nw, mblocks, nblocks, group_m, flags);
    ua2_count_launch(UA2_CNT_GEMM);
    if (split_flags) {
      const size_t total4 = (size_t)a.M * (a.N / 4);
      hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)std::min<size_t>((total4 + 255) / 256, 2048)), dim3(256), 0, s, a, ks);
    }
  };
  auto pick = [&](auto bmt_c) {
    if constexpr (kCanHo) { if (ho) { go(bmt_c, std::true_type{}, std::false_type{}); return; } }
    if (no_glds) go(bmt_c, std::false_type{}, std::false_type{});
    else go(bmt_c, std::false_type{}, std::true_type{});
  };
  if (bmt == 2) pick(std::integral_constant<int, 2>{});
  else if (bmt == 4) pick(std::integral_constant<int, 4>{});
  else pick(std::integral_constant<int, 8>{});
}

// Which of the two forms is faster — both give the same bits, so this is purely a cost model, fitted on
// tools/ubench/gemm_shapes.py (profiles/r1_gemm_shapes.txt):
//   skinny: the weights stream once per 64 rows at ~5 TB/s (+4 us per pass, +8 us for the two launches);
//   tiled : 1.1 us per two-chunk stage when the grid is small (latency-bound), ~700 TFLOP/s bf16 when large.
// UA2_SKINNY_MAX_ROWS=n overrides (skinny iff M <= n) for experiments.
bool choose_skinny(const ua2_linear_args& a, int nt) {
  const char* e = getenv("UA2_SKINNY_MAX_ROWS");   // read per call (launches are captured into graphs: not a per-frame cost)
  const int forced = e ? atoi(e) : -1;
  if (forced >= 0) return a.M <= forced;
  const double bytes = a.dtype == UA2_BF16 ? 2.0 : 4.0, kc = a.dtype == UA2_BF16 ? 32.0 : 16.0;
  const double w_bytes = (double)a.N * a.K * bytes * nt;
  const double t_skinny = 8.0 + (double)ua2_ceil_div(a.M, 16 * kSkinnyMT) * (w_bytes / 5.0e6 + 4.0);
  const double flop_rate = a.dtype == UA2_BF16 ? 700.0e6 : 45.0e6;    // flop per us
  const double t_tiled = std::max(5.0 + 1.1 * (a.K / kc) / kKS, 2.0 * a.M * a.N * a.K * nt / flop_rate);
  return t_skinny <= t_tiled;
}

// Rows up to which the weights-stationary kernel (ua2_skinny.hip) beats the tiled one on MI355X, per (N, K, matrices) of the model's
// Linear layers — both give the same bits, so this is a cost table, measured by tools/ubench/skinny_vs_tiled.py
// (profiles/r3_skinny_vs_tiled.txt); 320 for shapes it does not list (the cut-over of the round-3 sweeps).  The tiled kernel's time
// is nearly flat below ~256 rows (one workgroup per CU, ~0.16 us per chunk of K), the stationary kernel's grows with the operand
// bytes every CU has to pull in.
int skinny2_max_rows(int64_t N, int64_t K, int nt) {
  struct Cut { int N, K, nt, rows; };
  static constexpr Cut kCut[] = {{5120, 3072, 1, 176},  {3072, 3072, 1, 288}, {8192, 3072, 2, 96},  {3072, 8192, 1, 192}, {3072, 2048, 1, 288},
                                 {2048, 2048, 1, 320},  {8192, 2048, 2, 208}, {2048, 8192, 1, 320}, {2048, 3072, 1, 320}, {12296, 2048, 1, 192}};
  for (const Cut& c : kCut)
    if (c.N == N && c.K == K && c.nt == nt) return c.rows;
  return 320;
}

template <int DT>
int launch_dt(const ua2_linear_args& a, hipStream_t s, int force) {
  const int nt = a.epilogue == UA2_EPI_SWIGLU ? 2 : 1;
  const ua2_gemv_geometry geo = ua2_pick_gemv_geometry(a.dtype, a.N, a.K, nt);
  if (a.x_packed) {
    // the producer already wrote the operand in fragment order
  } else if (a.prologue == UA2_PRO_NORM) {
    launch_prep<DT, UA2_PRO_NORM>(a, geo.waves * 64, s);
  } else {
    launch_prep<DT, UA2_PRO_CAST>(a, geo.waves * 64, s);
  }
  UA2_LAUNCH_CHECK();
  // callers outside the row-invariance contract (ua2hip.h sum_order): the 256-row-tile kernel with one chain over K.
  bool order_free = a.sum_order == UA2_SUM_ORDER_FREE;
#ifdef UA2_GEMM_EXPERIMENTS
  // UA2_GEMM2_FORCE (experiment builds only: it breaks the LM's row-invariance contract): every eligible bf16 launch, whatever its
  // contract, for in-situ timing of LM prefill / big batches
  order_free = order_free || getenv("UA2_GEMM2_FORCE") != nullptr;
#endif
  if (order_free)
    if (const int rc = ua2_gemm2_try_launch(a, s); rc <= 0) return rc;
  UA2_CHECK(!a.y_ln_w, "ua2_linear: the LayerNorm hand-over (y_ln_w) is a form of the order-free kernel, which does not take this launch "
                       "(M=%d N=%d K=%d: ask ua2_linear_order_free_accepts first)", a.M, a.N, a.K);
  const bool skinny_ok = geo.waves * nt * kSkinnyMT * 1024 <= 128 * 1024;
  // The weights-stationary form (ua2_skinny.hip) serves the model's bf16 shapes up to a few hundred rows: measured against
  // the tiled kernel it wins up to 256 rows everywhere except the 128k-column lm_head (profiles/r3_skinny_sweep.txt).
  const bool prefer2 = force != 5 && a.dtype == UA2_BF16 && a.M <= skinny2_max_rows(a.N, a.K, nt) && a.N < 32768;
  const bool old_skinny = skinny_ok && (force == 4 || (force != 5 && choose_skinny(a, nt)));
  if (prefer2 || old_skinny)
    if (const int rc = ua2_skinny2_try_launch(a, geo, s); rc <= 0) return rc;
  if (old_skinny) {
    switch (a.epilogue) {
      case UA2_EPI_STORE: launch_skinny<DT, UA2_EPI_STORE>(a, geo, s); break;
      case UA2_EPI_RESIDUAL: launch_skinny<DT, UA2_EPI_RESIDUAL>(a, geo, s); break;
      case UA2_EPI_SWIGLU: launch_skinny<DT, UA2_EPI_SWIGLU>(a, geo, s); break;
      case UA2_EPI_QKV_ROPE: launch_skinny<DT, UA2_EPI_QKV_ROPE>(a, geo, s); break;
      case UA2_EPI_GELU: launch_skinny<DT, UA2_EPI_GELU>(a, geo, s); break;
      default: return 1;
    }
    UA2_LAUNCH_CHECK();
    return 0;
  }
  switch (a.epilogue) {
    case UA2_EPI_STORE: launch_gemm<DT, UA2_EPI_STORE>(a, geo.waves, s); break;
    case UA2_EPI_RESIDUAL: launch_gemm<DT, UA2_EPI_RESIDUAL>(a, geo.waves, s); break;
    case UA2_EPI_SWIGLU: launch_gemm<DT, UA2_EPI_SWIGLU>(a, geo.waves, s); break;
    case UA2_EPI_QKV_ROPE: launch_gemm<DT, UA2_EPI_QKV_ROPE>(a, geo.waves, s); break;
    case UA2_EPI_GELU: launch_gemm<DT, UA2_EPI_GELU>(a, geo.waves, s); break;
    default: return 1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" size_t ua2_linear_workspace_bytes(int dtype, int64_t M, int64_t K) {
  if (M <= 0 || K <= 0 || (dtype != UA2_BF16 && dtype != UA2_F32)) return 0;
  const int kc = dtype == UA2_BF16 ? 32 : 16;
  return (size_t)((M + 15) / 16) * (size_t)((K + kc - 1) / kc) * 1024;
}

int ua2_gemm_try_launch(const ua2_linear_args& a, hipStream_t s, int force) {
  if (a.prologue != UA2_PRO_CAST && a.prologue != UA2_PRO_NORM && !(a.prologue == UA2_PRO_SCALED && a.x_packed)) return 1;
  if (!a.x_packed && (!a.workspace || a.workspace_bytes < ua2_linear_workspace_bytes(a.dtype, a.M, a.K))) return 1;
  const int rt = ua2_gemv_rows_per_tile(a.dtype, a.K);
  if (rt < 1) return 1;                          // the decode kernel cannot take this K at all: nothing to be identical with
  if (!force && a.M <= ua2_gemv_rows_preferred(a.dtype, a.K)) return 1;   // a few rows: the decode kernel (operand rows live in LDS); up to `rt` rows it COULD (forced mode 2 / row-major hand-overs)
  if (a.dtype == UA2_BF16) return launch_dt<UA2_BF16>(a, s, force);
  return launch_dt<UA2_F32>(a, s, force);
}
