// Prototype (measurement only, not part of libua2hip.so): ONE depth-decoder pass at B = 1 — 4 layers x (qkv, o, SwiGLU, down)
// + audio head, 536 MB of bf16 weights — as ONE persistent launch on the loader / consumer engine that
// /opt/skills/guides/MI355X_MICROARCH.md prices (rows launches-baseline / engine-vs-launches / ldsdma-fill / allgather):
//   * 256 workgroups (one per CU, all resident) x 7 waves: 1 LDS-DMA LOADER (non-temporal weight stream into a ring of 8 x 16 KiB
//     slots, running ahead of the dependency chain by up to the ring; -DENG_LOADERS=2 for two), 3 CONSUMERS (MFMA chains over
//     the slots they own, per-range partial sums through LDS; the wave that owns a unit's last slot adds them in range order,
//     runs the epilogue and publishes), 2 + 1 GATHER waves (sweep the previous op's output granules into the LDS operand buffer;
//     the third one turns the sum-of-squares partials into the row scale);
//   * no grid barrier: an op's output reaches every CU as 8-byte {tag, data} granules (one sc1 store each; the data IS the
//     flag), cdna_hip_programming.md Guideline 16 recipe R2; granule arrays are per op, zeroed by a memset in front of the launch;
//   * the unit of work is 8 output columns (half an MFMA tile; the weights are re-packed [unit][K/32][4][8][16 B] so that a
//     unit's bytes are contiguous): N = 2048 ops have 256 units = one per CU, so the fp32 residual stream never leaves its CU;
//   * the arithmetic is ua2_linear's B = 1 scaled plan (csrc/ua2_gemv.hip + linear_epilogue): one MFMA chain from zero per
//     K range (`ranges` = the decode kernel's wave count for the shape), partial sums added in range order, y = rstd * sum with
//     rstd from the producer's sum-of-squares partials in scaled_rstd_reduce's order, hand-over h = RNE_bf16(out * w_next):
//     the outputs are compared BIT FOR BIT with the same chain through ua2_linear (tools/ubench/engine_run.py).
// Stand-in: the depth decoder's attention (<= 8 cached positions; at step 0 softmax over one key) is replaced by "o-proj
// consumes bf16(q)" on both sides — same bytes on the edge (8 KB), no KV reads (16 KB per layer in production).
// Every spin is bounded (wall clock); a give-up code lands in err[0] and every wave of every workgroup leaves.
// RESULT (profiles/r5_engine_prototype.txt, r5_notes.md §8): bit-identical to the chain on every op; 150-152 us per pass against
// 142 us for the 17 launches as one graph (1.05-1.07x: not faster).  What bounds it: a gather pass is a vector load queued
// behind the CU's own weight requests (4 us per pass with 96 KiB in flight), 17 all-to-all edges per pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ua2_common.h"
#include "ua2_linear_common.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

#define ENG_MAX_OPS 20
#define ENG_SLOTS 8
#define ENG_SLOT_BYTES 16384

struct eng_op {
  const char* w0;
  const char* w1;        // SwiGLU: up matrix
  const float* nw;       // kind 1: weight of the consuming RMSNorm (hand-over)
  u64* gin;              // operand granules: K / 2 bf16 pairs
  u64* gssq_in;          // kinds 0 / 2: K / 8 sum-of-squares partials (one per 8 columns of the producer)
  u64* gout;             // published bf16 pairs (kind 0: the first pub_n outputs; kind 1: hand-over; kind 2: activations)
  u64* gssq_out;         // kind 1: N / 8 partials
  float* y;              // fp32 result, plain stores (read by the host after the launch)
  int N, K, ranges, kind;   // kind: 0 = STORE (scaled in), 1 = RESIDUAL (cast in, hand-over out), 2 = SWIGLU (scaled in)
  int xsel, pub_n;       // LDS operand buffer (0 / 1: 4 KiB, 2: 16 KiB); kind 0: how many outputs are published
  float eps;
  int pad;
};
struct eng_args {
  eng_op op[ENG_MAX_OPS];
  const float* x0;       // residual stream on entry [2048]
  unsigned* err;         // [0] give-up code (0 = none), [1..] scratch
  long long* stamps;     // NULL or [ncu][64] wall-clock ticks (100 MHz): per op p [3p] start, [3p+1] operand ready, [3p+2] done;
                         // [60] consumer's ticks waiting for landed slots, [61] loader's ticks waiting for free slots, [62] start
  int nops, ncu;
  int timeout_ticks;     // wall_clock64 ticks (100 MHz)
  int flags;             // bit 0: thin the loader while this CU gathers; debug: 2 no loader, 4 no consumers, 8 no gather waves (the rest runs into its time-out);
                         // timing knock-outs (wrong results): 16 consumers do not wait for their operand, 32 gather waves do not wait for tags, 64 the loader does not wait for free slots,
                         // 128 no epilogues (nothing stored or published), 256 no fragment reads / MFMAs
  int evt_op, pad;       // stamps: the op whose consumer events CU 0 records
};

namespace {

constexpr int kRing = 0, kXA = ENG_SLOTS * ENG_SLOT_BYTES, kXB0 = kXA + 16384, kXB1 = kXB0 + 4096, kMisc = kXB1 + 4096;
constexpr int kPart = kMisc + 2048, kLds = kPart + 4096;
// misc (ints): [48..49] slots landed per loader wave, [2],[3] operand ready (op index + 1) per gather wave, [4] abort, [50..51] a gather wave is sweeping (loader thinning),
//              [8..9] rstd per op parity, [16..23] ring slot freed (sequence + 1), [24..31] residual stream (8 floats),
//              [32..47] partials of slot s written (s + 1), [64..319] staged sum-of-squares partials
// part: [16 slots][8 ranges][8 columns] floats

__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// LDS flags order LDS traffic only: wait for this wave's LDS operations, then a relaxed store.  (A workgroup-scope RELEASE also waits
// for the wave's outstanding GLOBAL operations — vmcnt(0) — i.e. for the write-through stores of the epilogue it has just issued.)
__device__ __forceinline__ void lds_st(int* p, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Clock {
  long long t0; int limit;
  __device__ __forceinline__ bool expired() const { return (long long)wall_clock64() - t0 > limit; }
};

// wave-uniform spin on an LDS word; false = give up (abort flag seen or out of time)
template <class Pred>
__device__ __forceinline__ bool spin(int* misc, const Clock& ck, unsigned* err, unsigned code, Pred&& pred, long long* waited = nullptr) {
  if (pred()) return true;
  const long long w0 = waited ? (long long)wall_clock64() : 0;
  for (unsigned it = 0;; ++it) {
    if (pred()) { if (waited) *waited += (long long)wall_clock64() - w0; return true; }
    if ((it & 63u) == 63u) {
      if (lds_ld(misc + 4) != 0) return false;
      if (ck.expired()) {
        lds_st(misc + 4, 1);
        if ((threadIdx.x & 63) == 0) atomicCAS(err, 0u, code);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p); }

// four 1 KiB LDS-DMA requests (16 B per lane), non-temporal: block j from s + 1024 j + voff to LDS d + 1024 j (+ 16 lane)
#ifndef ENG_M0_ADD
#define ENG_M0_ADD 0      // the instruction offset moves the LDS address as well as the global one (M0 + offset + 16 lane): M0 stays put
#endif
__device__ __forceinline__ void dma4(const char* s_, unsigned voff, unsigned d_) {
  // wave-uniform by construction; said explicitly (under SGPR pressure the compiler keeps uniform values in VGPRs)
  const unsigned d = __builtin_amdgcn_readfirstlane(d_);
  const uintptr_t sa = (uintptr_t)s_;
  // (the builtin returns int: without the unsigned casts a low word >= 2^31 sign-extends over the high word)
  const char* s = (const char*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32) |
                                (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa));
  unsigned keep;
#if ENG_M0_ADD
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 nt\n\t"
      "s_add_u32 m0, %3, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024 nt\n\t"
      "s_add_u32 m0, %3, 2048\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:2048 nt\n\t"
      "s_add_u32 m0, %3, 3072\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:3072 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(s), "v"(voff), "s"(d)
      : "memory", "scc");
#else
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %1 nt\n\t"
      "global_load_lds_dwordx4 %2, %1 offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %2, %1 offset:2048 nt\n\t"
      "global_load_lds_dwordx4 %2, %1 offset:3072 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(s), "v"(voff), "s"(d)
      : "memory");
#endif
}

__device__ __forceinline__ int units_of(const eng_op& o) { return o.N >> 3; }

// ---------------------------------------------------------------- loader
// NL loader waves: wave lw takes the slots s = lw (mod NL) of the CU's slot sequence (a wave has 63 requests = 63 KiB in flight at most)
template <int NL>
__device__ __forceinline__ void loader(const eng_args& a, char* smem, int* misc, const Clock& ck, int cu, int lw, int lane) {
  const unsigned ring = lds_addr(smem + kRing);
  const unsigned voff = (unsigned)lane * 16u;
  const bool thin = a.flags & 1;
  int s = 0, mine = 0;
  long long waited = 0;
  for (int p = 0; p < a.nops; ++p) {
    const eng_op& o = a.op[p];
    const int nm = o.kind == 2 ? 2 : 1;
    const int nsl = o.K >> 10;                                  // slots per unit: K * 16 B / 16 KiB
    for (int u = cu; u < units_of(o); u += a.ncu)
      for (int m = 0; m < nm; ++m) {
        const char* src = (m ? o.w1 : o.w0) + (size_t)u * ((size_t)o.K * 16);
        for (int i = 0; i < nsl; ++i, ++s) {
          if (s % NL != lw) continue;
          if (s >= ENG_SLOTS && !(a.flags & 64))
            if (!spin(misc, ck, a.err, 0x100u + p, [&] { return lds_ld(misc + 16 + (s & (ENG_SLOTS - 1))) == s - ENG_SLOTS + 1; }, a.stamps ? &waited : nullptr)) return;
          const unsigned d = ring + (unsigned)(s & (ENG_SLOTS - 1)) * ENG_SLOT_BYTES;
          const char* sp = src + (size_t)i * ENG_SLOT_BYTES;
#pragma unroll
          for (int j = 0; j < 4; ++j) dma4(sp + j * 4096, voff, d + j * 4096);
          ++mine;
          // at most 3 of this wave's slots (48 requests) in flight; a wave's requests land in order: all but its last two slots have landed
          if (thin && (lds_ld(misc + 50) | lds_ld(misc + 51))) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_st(misc + 48 + lw, mine);
          } else {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            if (mine >= 2) lds_st(misc + 48 + lw, mine - 2);
          }
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_st(misc + 48 + lw, mine);
  if (a.stamps && lane == 0 && lw == 0) a.stamps[(size_t)cu * 64 + 61] = waited;
}

// ---------------------------------------------------------------- gather
__device__ __forceinline__ u64 gran_ld(const u64* p) { return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gran_st(u64* p, unsigned v) {
  __hip_atomic_store((gu64*)p, (1ull << 32) | (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one block of 64 x NL granules, re-read until every tag is set; values -> dst[i] (LDS)
template <int NL>
__device__ __forceinline__ bool sweep(const u64* g, unsigned* dst, int lane, int* misc, const Clock& ck, unsigned* err, unsigned code, bool nowait,
                                      int* thin = nullptr) {
  unsigned v[NL];
  if (nowait) {                                                 // timing knock-out (wrong results): one pass, tags ignored
#pragma unroll
    for (int k = 0; k < NL; ++k) dst[lane + 64 * k] = (unsigned)gran_ld(g + lane + 64 * k);
    return true;
  }
#ifndef ENG_NO_SAMPLE_POLL
  // poll ONE granule until it is there, then sweep: 256 CUs x 3 waves re-reading 4 KiB each per pass is TB/s of traffic next to the
  // weight stream (MI355X_MICROARCH.md polling-cost); the producers finish within a microsecond or two of each other
  for (unsigned it = 0;; ++it) {
    const u64 x = gran_ld(g + (lane & 7) * (NL * 8));            // 8 distinct granules spread over the block, 8 lanes each
    if (__all((x >> 32) == 1ull)) break;
    if ((it & 15u) == 15u) {
      if (lds_ld(misc + 4) != 0) return false;
      if (ck.expired()) {
        lds_st(misc + 4, 1);
        if (lane == 0) atomicCAS(err, 0u, code);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(2);
  }
#endif
  if (thin) lds_st(thin, 1);
  for (unsigned it = 0;; ++it) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const u64 x = gran_ld(g + lane + 64 * k);
      v[k] = (unsigned)x;
      ok &= (x >> 32) == 1ull;
    }
    if (__all(ok)) break;
    if ((it & 15u) == 15u) {
      if (lds_ld(misc + 4) != 0) return false;
      if (ck.expired()) {
        lds_st(misc + 4, 1);
        if (lane == 0) atomicCAS(err, 0u, code);
        return false;
      }
    }
#ifdef ENG_SWEEP_SLEEP
    __builtin_amdgcn_s_sleep(ENG_SWEEP_SLEEP);
#endif
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) dst[lane + 64 * k] = v[k];
  return true;
}

__device__ __forceinline__ void gather(const eng_args& a, char* smem, int* misc, const Clock& ck, int g, int lane) {
  for (int p = 0; p < a.nops; ++p) {
    const eng_op& o = a.op[p];
    if (g < 2) {
      unsigned* xb = reinterpret_cast<unsigned*>(smem + (o.xsel == 2 ? kXA : (o.xsel == 1 ? kXB1 : kXB0)));
      const int n = o.K >> 1, half = n >> 1;                      // granules; this wave's half in blocks of 512 (a finished block is not read again;
                                                                  // all 2048 of a K = 8192 operand per pass was measured: 149.8 -> 168 us per pass of the decoder)
      int* thin = (a.flags & 1) ? misc + 50 + g : nullptr;        // loader thinning: only while the sweep itself runs (after the sample has arrived)
      for (int b = 0; b < half; b += 512)
        if (!sweep<8>(o.gin + g * half + b, xb + g * half + b, lane, misc, ck, a.err, 0x200u + p, a.flags & 32, thin)) return;
      if (thin) lds_st(thin, 0);
      lds_st(misc + 2 + g, p + 1);
    } else {
      if (o.kind != 1) {
        // row scale: K / 8 half-tile partials -> K / 16 tile partials (lo + hi: the last level of ssq_tile16) -> scaled_rstd_reduce's order
        float* st = reinterpret_cast<float*>(misc + 64);
        if (!sweep<4>(o.gssq_in, reinterpret_cast<unsigned*>(st), lane, misc, ck, a.err, 0x300u + p, a.flags & 32)) return;   // K = 2048: 256 partials
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const int c = lane & 15, nparts = o.K >> 4;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = c + 16 * i;
          const float part = (t < nparts) ? __fadd_rn(st[2 * t], st[2 * t + 1]) : 0.f;
          s = __fadd_rn(s, part);
        }
        s = __fadd_rn(s, __shfl_xor(s, 1));
        s = __fadd_rn(s, __shfl_xor(s, 2));
        s = __fadd_rn(s, __shfl_xor(s, 4));
        s = __fadd_rn(s, __shfl_xor(s, 8));
        const float rstd = 1.0f / sqrtf(s / (float)o.K + o.eps);
        if (lane == 0) reinterpret_cast<float*>(misc + 8)[p & 1] = rstd;
      }
      lds_st(misc + 6, p + 1);
    }
  }
}

// ---------------------------------------------------------------- consumers
// NC consumer waves: wave w takes the slots s = w (mod NC) of the CU's slot sequence (every wave walks the same (op, unit, matrix,
// slot) enumeration).  A slot's per-range partial sums go to LDS (part[s % 16][range][column]); the wave that owns a unit's LAST
// slot waits for the others' partials, adds them in range order (the decode kernel's order) and runs the epilogue.
//
// One MFMA takes TWO half-chunks of the unit: columns 0-7 of B hold half-chunk hc (range r), columns 8-15 half-chunk hc' (range
// r + RPS / 2); row 0 of A holds x[hc], row 1 x[hc'].  D[0][0..7] is range r's chain, D[1][8..15] range r + RPS/2's — each
// element the same k-sum, in the same order, as the decode kernel's (the cross terms land in elements nobody reads).  Every lane
// of a B read fetches live bytes and the LDS reads per slot halve (they, not the MFMAs, bound a consumer: 64 x 1 KiB per slot before).
template <int CPR>
__device__ __forceinline__ void slot_partials(const char* slot, const char* xk, float* pp, int lane) {
  constexpr int RPS = 32 / CPR, HR = RPS / 2;
  const bool hi = (lane & 8) != 0;
  const char* base = slot + ((lane >> 4) * 8 + (lane & 7)) * 16 + (hi ? HR * CPR * 512 : 0);
  const char* xa = xk + (lane >> 4) * 16 + ((lane & 15) == 1 ? HR * CPR * 64 : 0);
  f32x4 acc[HR];
#pragma unroll
  for (int r = 0; r < HR; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < CPR; ++c)
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const int hc = r * CPR + c;
      const u32x4 w = *reinterpret_cast<const u32x4*>(base + hc * 512);
      const u32x4 x = *reinterpret_cast<const u32x4*>(xa + hc * 64);
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), acc[r], 0, 0, 0);
    }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < HR; ++r) pp[((hi ? HR : 0) + r) * 8 + (lane & 7)] = hi ? acc[r][1] : acc[r][0];
  }
}

// bf16 pairs of 8 consecutive columns (lanes 0-7 and their copies in 8-15): lane j holds column n0 + j
__device__ __forceinline__ void publish_pairs(u64* g, int n0, int lane, float v) {
  const unsigned lo = f2bf(v);
  const unsigned hi = __shfl_down(lo, 1);
  if (lane < 8 && (lane & 1) == 0) gran_st(g + ((n0 + lane) >> 1), lo | (hi << 16));
}

template <int NC, int kLoadersC>
__device__ __forceinline__ void consumer(const eng_args& a, char* smem, int* misc, const Clock& ck, int cu, int w, int lane) {
  int s = 0;
  const int j = lane & 7;
  float* part = reinterpret_cast<float*>(smem + kPart);
  float* resid = reinterpret_cast<float*>(misc + 24);
  ua2_linear_args dummy;
  dummy.act_kind = 0;
  long long waited = 0;
  long long* wp = (a.stamps && w == 0) ? &waited : nullptr;
  long long* st = a.stamps ? a.stamps + (size_t)cu * 64 : nullptr;        // start / ready: the first wave's; done: the last wave's
  if (st && lane == 0 && w == 0) st[62] = ck.t0;
  // events of CU 0 in op ENG_EVT_OP: [256 * 64 + 96 w + k] = (kind << 56) | ticks since launch; kinds: 1 slot landed, 2 slot done, 3 partials of the unit in, 4 epilogue done
  long long* ev = (a.stamps && cu == 0) ? a.stamps + 256 * 64 + 96 * w : nullptr;
  int nev = 0;
  auto evt = [&](int p, long long kind) {
    if (ev && p == a.evt_op && nev < 96 && lane == 0) ev[nev++] = (kind << 56) | ((long long)wall_clock64() - ck.t0);
  };
  for (int p = 0; p < a.nops; ++p) {
    const eng_op& o = a.op[p];
    if (st && lane == 0) atomicMin(reinterpret_cast<unsigned long long*>(st + 3 * p), (unsigned long long)wall_clock64());
    if (!(a.flags & 16))
      if (!spin(misc, ck, a.err, 0x400u + p, [&] { return lds_ld(misc + 2) > p && lds_ld(misc + 3) > p && lds_ld(misc + 6) > p; })) return;
    if (st && lane == 0) atomicMin(reinterpret_cast<unsigned long long*>(st + 3 * p + 1), (unsigned long long)wall_clock64());
    const char* xb = smem + (o.xsel == 2 ? kXA : (o.xsel == 1 ? kXB1 : kXB0));
    const float rstd = reinterpret_cast<const float*>(misc + 8)[p & 1];
    const int nsl = o.K >> 10, cpr = (o.K >> 5) / o.ranges, rps = 32 / cpr;
    const int nm = o.kind == 2 ? 2 : 1;
    // the hand-over's norm weight: requested before the slots (a cold load in the epilogue: 1.3 us on every edge behind a RESIDUAL op)
    const float nwv = (o.kind == 1 && o.nw) ? o.nw[cu * 8 + j] : 0.f;
    for (int u = cu; u < units_of(o); u += a.ncu) {
      const int s0 = s;
      for (int m = 0; m < nm; ++m)
        for (int i = 0; i < nsl; ++i, ++s) {
          if (s % NC != w) continue;
          if (!spin(misc, ck, a.err, 0x500u + p, [&] { return lds_ld(misc + 48 + s % kLoadersC) > s / kLoadersC; }, wp)) return;
          evt(p, 1);
          const char* slot = smem + kRing + (s & (ENG_SLOTS - 1)) * ENG_SLOT_BYTES;
          const char* xk = xb + (size_t)i * 32 * 64;
          float* pp = part + (s & 15) * 64;
          if (a.flags & 256) {}                                  // knock-out: no fragment reads, no MFMAs
          else if (cpr == 4) slot_partials<4>(slot, xk, pp, lane);
          else if (cpr == 8) slot_partials<8>(slot, xk, pp, lane);
          else slot_partials<16>(slot, xk, pp, lane);
          lds_st(misc + 32 + (s & 15), s + 1);                 // partials written (release: also this wave's reads of the slot are done)
          lds_st(misc + 16 + (s & (ENG_SLOTS - 1)), s + 1);    // the ring slot is free
          evt(p, 2);
        }
      if ((s - 1) % NC != w) continue;                         // the unit's last slot was not this wave's: somebody else finishes it
      if (!spin(misc, ck, a.err, 0x600u + p, [&] {
            bool ok = true;
            for (int t = s0; t < s - 1; ++t) ok &= lds_ld(misc + 32 + (t & 15)) == t + 1;
            return ok;
          })) return;
      evt(p, 3);
      float tot[2];
      const int rsh = rps == 8 ? 3 : (rps == 4 ? 2 : 1);
      for (int m = 0; m < nm; ++m) {
        float pv[16];                                            // every partial requested before the first add (a dependent LDS round trip each otherwise)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int qq = min(q, o.ranges - 1);
          pv[q] = part[((s0 + m * nsl + (qq >> rsh)) & 15) * 64 + (qq & (rps - 1)) * 8 + j];
        }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (q < o.ranges) sum += pv[q];                        // range order
        tot[m] = sum;
      }
      const int n = u * 8 + j;
      if (a.flags & 128) { evt(p, 4); continue; }                // knock-out: no epilogue
      if (o.kind == 0) {
        const float v = __fmul_rn(tot[0], rstd);
        if (lane < 8) o.y[n] = v;
        if (u * 8 < o.pub_n) publish_pairs(o.gout, u * 8, lane, v);
      } else if (o.kind == 2) {
        const float out = ua2_act_glu(dummy, __fmul_rn(tot[0], rstd), __fmul_rn(tot[1], rstd));
        if (lane < 8) o.y[n] = out;
        publish_pairs(o.gout, u * 8, lane, out);
      } else {
        const float out = __fadd_rn(tot[0], resid[j]);
        if (lane < 8) { resid[j] = out; o.y[n] = out; }
        if (o.nw) {
          float sq = __fmaf_rn(out, out, __shfl_xor(__fmul_rn(out, out), 1));     // ssq_tile16's first three levels over this unit's 8 columns
          sq = __fadd_rn(sq, __shfl_xor(sq, 2));
          sq = __fadd_rn(sq, __shfl_xor(sq, 4));
          if (lane == 0) gran_st(o.gssq_out + u, __float_as_uint(sq));
          publish_pairs(o.gout, u * 8, lane, __fmul_rn(out, nwv));   // RESIDUAL ops have one unit per CU (N = 8 x ncu)
        }
      }
      evt(p, 4);
    }
    if (st && lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(st + 3 * p + 2), (unsigned long long)wall_clock64());
  }
  if (st && lane == 0 && w == 0) st[60] = waited;
}

#ifndef ENG_LOADERS
#define ENG_LOADERS 1
#endif
constexpr int kConsumers = 3, kLoaders = ENG_LOADERS;

__global__ __launch_bounds__((3 + kLoaders + kConsumers) * 64, 1) void engine_kernel(const eng_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* misc = reinterpret_cast<int*>(smem + kMisc);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 64) misc[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < 8) reinterpret_cast<float*>(misc + 24)[threadIdx.x] = a.x0[blockIdx.x * 8 + threadIdx.x];
  __syncthreads();
  Clock ck{(long long)wall_clock64(), a.timeout_ticks};
  const int cu = blockIdx.x;
  if (wave < kLoaders) { if (!(a.flags & 2)) loader<kLoaders>(a, smem, misc, ck, cu, wave, lane); }
  else if (wave < kLoaders + kConsumers) { if (!(a.flags & 4)) consumer<kConsumers, kLoaders>(a, smem, misc, ck, cu, wave - kLoaders, lane); }
  else if (!(a.flags & 8)) gather(a, smem, misc, ck, wave - kLoaders - kConsumers, lane);
}

}  // namespace

extern "C" int eng_lds_bytes() { return kLds; }

// zero [zero, zero + zero_bytes) (every granule array of the pass except the host-written entry edge), then the launch
extern "C" int eng_launch(const eng_args* a, void* zero, size_t zero_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(engine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return -1;
    once = true;
  }
  if (zero_bytes && hipMemsetAsync(zero, 0, zero_bytes, s) != hipSuccess) return -2;
  hipLaunchKernelGGL(engine_kernel, dim3(a->ncu), dim3((3 + kLoaders + kConsumers) * 64), kLds, s, *a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// average milliseconds of `iters` back-to-back passes (memset + launch each), HIP events on the stream
extern "C" int eng_timed(const eng_args* a, void* zero, size_t zero_bytes, int iters, void* stream, float* ms) {
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i)
    if (int rc = eng_launch(a, zero, zero_bytes, stream)) return rc;
  (void)hipEventRecord(e1, s);
  if (hipEventSynchronize(e1) != hipSuccess) return -4;
  (void)hipEventElapsedTime(ms, e0, e1);
  *ms /= (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}
