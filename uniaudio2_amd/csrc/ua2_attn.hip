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
#include <stdlib.h>

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

// kG = query heads per kv head, a template parameter: with a run-time G padded to kMaxG = 4 the Llama trunk (G = 3) spent a
// quarter of its vector-ALU work on a head that does not exist (the launch is ALU-bound from 64 rows up, profiles/r3_notes.md)
// PF (round 6): true = K / V of step t + 1 requested before the arithmetic of step t (the B = 1 form: one workgroup per CU, latency
// is everything); false = loads at the top of their own step and the kernel capped at 128 registers (bf16, HS = 128, G = 3 needs 164
// with the prefetch), so that TWO workgroups share a CU — batched decode launches more workgroups than the device has CUs (64 rows x 8
// kv heads = 512, 1024 rows = 8192) and each of them is a chain of dependent round trips (position -> page id -> K / V -> states
// through LDS): occupancy, not lookahead, is what such a launch waits for.  Same arithmetic, same order: same bits.
template <int DT, int HS, int kG, bool PF = true>
__global__ __launch_bounds__(kFusedWaves * 64, PF ? 2 : 4) void attn_fused_kernel(const ua2_attn_args a) {
  constexpr int EPL = Elem<DT>::EPL, BYTES = Elem<DT>::BYTES;
  constexpr int LPR = HS / EPL, RPW = 64 / LPR;   // lanes per cache row, row groups per wave
  constexpr int UNR = (DT == UA2_BF16) ? 4 : 2;   // wave instructions per step (K and V each)
  constexpr int NS = kFusedWaves * RPW;           // independent online-softmax states per workgroup
  // Every 16-/8-lane row group keeps its own (m, l, o) state: no cross-group traffic inside the
  // loop (the first version spent ~6 us in serialized ds_bpermute chains).  States merge once,
  // through LDS, in (wave, group) order.
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int r = blockIdx.x, kvh = blockIdx.y;
  constexpr int G = kG;
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

  float q[kG][EPL];
#pragma unroll
  for (int h = 0; h < kG; ++h) {
    const float* qp = a.q + ((size_t)r * a.kv.n_head + (size_t)kvh * G + h) * HS + sub * EPL;
    const float sc = scale;
#pragma unroll
    for (int e4 = 0; e4 < EPL / 4; ++e4) {
      const float4 t = *reinterpret_cast<const float4*>(qp + 4 * e4);
      q[h][4 * e4 + 0] = t.x * sc; q[h][4 * e4 + 1] = t.y * sc; q[h][4 * e4 + 2] = t.z * sc; q[h][4 * e4 + 3] = t.w * sc;
    }
  }
  float m_run[kG], l_run[kG], o_run[kG][EPL];
#pragma unroll
  for (int h = 0; h < kG; ++h) {
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
      const size_t off = ((((size_t)ptab[ua2_page_slot(a.kv, jc)] * a.kv.n_kv + kvh) * UA2_PAGE + (jc % UA2_PAGE)) * HS +
                          (size_t)sub * EPL) * BYTES;
      kr[u] = *reinterpret_cast<const u32x4*>((const char*)a.kv.k_pool + off);
      vr[u] = *reinterpret_cast<const u32x4*>((const char*)a.kv.v_pool + off);
    }
  };
  u32x4 kraw[UNR], vraw[UNR], knext[PF ? UNR : 1], vnext[PF ? UNR : 1];
  if constexpr (PF) { if (j0 < j1) issue(j0, kraw, vraw); }
  for (int jb = j0; jb < j1; jb += UNR * RPW) {
    const bool more = PF && jb + UNR * RPW < j1;
    if constexpr (PF) { if (more) issue(jb + UNR * RPW, knext, vnext); }
    else issue(jb, kraw, vraw);
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) ok[u] = (jb + u * RPW + rin) < j1;
    float s[UNR][kG];
    float gmax[kG];
#pragma unroll
    for (int h = 0; h < kG; ++h) gmax[h] = -INFINITY;
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
      for (int h = 0; h < kG; ++h) {
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
    for (int h = 0; h < kG; ++h) {
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
      for (int h = 0; h < kG; ++h) {
        const float p = ok[u] ? fast_exp(s[u][h] - m_run[h]) : 0.f;
        l_run[h] += p;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o_run[h][e] += p * vf[e];
      }
    }
    if constexpr (PF) {
      if (more) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) { kraw[u] = knext[u]; vraw[u] = vnext[u]; }
      }
    }
  }
  // publish the state of this row group
  const int sidx = wave * RPW + rin;
#pragma unroll
  for (int h = 0; h < kG; ++h) {
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

template <int DT, int HS, int kG>
void launch_fused_g(const ua2_attn_args& a, hipStream_t s) {
  constexpr int EPL = Elem<DT>::EPL, RPW = 64 / (HS / EPL), NS = kFusedWaves * RPW;
  const size_t smem = (size_t)(2 * NS * kMaxG + (size_t)NS * kG * HS) * sizeof(float);
  // more workgroups than CUs: the two-per-CU form (bf16: the fp32 kernel already fits twice)
  static const int dense_env = getenv("UA2_ATTN_DENSE") ? atoi(getenv("UA2_ATTN_DENSE")) : -1;      // A/B (read once): 0 = never, 1 = always
  const bool dense = DT == UA2_BF16 && (dense_env >= 0 ? dense_env != 0 : (int64_t)a.R * a.kv.n_kv > 256);
  if (dense) {
    if constexpr (DT == UA2_BF16 && kG <= 3) {          // four query heads per kv head do not fit 128 registers (80-164 B of scratch)
      constexpr auto kern2 = attn_fused_kernel<DT, HS, kG, false>;
      ua2_allow_big_lds<kern2>();
      hipLaunchKernelGGL(kern2, dim3(a.R, a.kv.n_kv), dim3(kFusedWaves * 64), smem, s, a);
      return;
    }
  }
  constexpr auto kern = attn_fused_kernel<DT, HS, kG>;
  ua2_allow_big_lds<kern>();
  hipLaunchKernelGGL(kern, dim3(a.R, a.kv.n_kv), dim3(kFusedWaves * 64), smem, s, a);
}

template <int DT, int HS>
void launch_fused_hs(const ua2_attn_args& a, hipStream_t s) {
  switch (a.kv.n_head / a.kv.n_kv) {       // <= kMaxG (checked by ua2_attn_launch)
    case 1: launch_fused_g<DT, HS, 1>(a, s); break;
    case 2: launch_fused_g<DT, HS, 2>(a, s); break;
    case 3: launch_fused_g<DT, HS, 3>(a, s); break;
    default: launch_fused_g<DT, HS, 4>(a, s); break;
  }
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

// ---- many query rows per sequence: MFMA flash attention (prefill, dense encoders / DiT) -------------------------
// north_star: "MFMA ... with LDS-staged KV tiles".  The row-by-row kernel above re-reads a sequence's K/V once per
// query row (258 us per layer at 2048 prefill rows, 1.2 ms at 32 x 195, profiles/r2_*); here a workgroup takes up to
// QT x 16 query rows of ONE sequence for one kv head, stages each 64-position K / V page of that head in LDS once
// (the page is contiguous in the pool), and its QT x G waves — one per (16-row tile, query head of the group) — run
//     S^T = K Q^T      v_mfma_f32_16x16x32_bf16, A = K rows from LDS, B = the wave's Q tile (registers)
//     online softmax   per query row = per lane column (q = lane & 15); 16 local values + two cross-group shuffles
//     O^T += V^T P^T   A = V^T from LDS (V is transposed by the staging writes), B = P straight from the S^T registers
// The transposed formulation keeps everything "query = lane & 15"-major: the S^T accumulator registers ARE the B operand
// of the second product (its reduction index is simply enumerated as the accumulator holds the keys: 4 + 4 per 32-key
// chunk, which V^T is read to match), so P never travels through LDS, and the softmax rescale is lane-local.
// Numerics (bf16 contract, DESIGN.md §2: K/V bf16, everything else fp32): q (pre-scaled by log2(e)/sqrt(hs)) and p are
// split into bf16 hi + lo halves (16 significant bits, two MFMAs each), accumulation fp32 — fp32-grade scores and
// weights on bf16 keys and values, the contract the row-by-row kernel and the oracle implement.
// A row's result is a function of its own q, its position and the cache: key blocks are visited in order 0, 1, ...,
// masked keys contribute exact zeros, MFMA output rows do not see each other — the composition of tiles and groups never
// changes a row's bits (tests/test_gpu_invariance.py).
// SPLIT = false (ua2_attn_args.flags & UA2_ATTN_BF16_QP: callers outside the fp32-grade-softmax contract, i.e. the codec's DiT,
// whose reference runs torch SDPA under bf16 autocast — q, k, v AND the softmax weights in bf16, reason_tokenizer.py:265): q and p are
// rounded to bf16 once (no lo halves): half the MFMAs of both products and none of the lo-half conversions.
// NPG = 2 (the DiT's instantiation, order-free callers only): TWO pages = 128 keys per loop iteration — one online-softmax step (a max
// and a sum across the lane groups, one rescale of O) and one workgroup barrier per 128 keys instead of per 64, 16 independent MFMAs
// per product instead of 8.  A block's time is its dependent chain (barrier -> K fragments -> S^T -> two shuffles -> exp2 -> two
// shuffles -> V^T -> O), not its arithmetic: 8 iterations of ~1.9 us at 500 keys (profiles/r6_notes.md §16).  The softmax steps see
// other block boundaries, so the bits differ from NPG = 1 (fp32 rounding of the rescales): not for callers under the row-invariance
// contract, whose rows must not depend on how the keys are blocked.
template <int HS, int G, int QT, bool SPLIT = true, int NPG = 1>
__global__ __launch_bounds__(64 * G * QT) void attn_flash_kernel(const ua2_attn_args a) {
  constexpr int NW = G * QT;
  constexpr int KPI = NPG * UA2_PAGE;         // keys per loop iteration
  constexpr int DC = HS / 32;                 // 32-dim chunks of the QK product
  constexpr int DTL = HS / 16;                // 16-dim tiles of the output
  constexpr int KROW = HS * 2 + 16;           // bytes per key row of the K image (pad: conflict-free 16-byte reads)
  // V image: the page as it lies in the pool, [64 keys][HS dims] — a 16-byte copy per thread and piece; the V^T operand of the second
  // product comes out of ds_read_b64_tr_b16 (gfx950's transposing LDS read: each 16-lane group reads a [4 keys][16 dims] block
  // through per-lane 8-byte addresses and lane ql receives dim ql's four keys).  Round 6: before, the staging writes transposed —
  // 16 ds_write_b16 per thread and key block, most of a block's LDS time (profiles/r5_notes.md §5).  Row pitch = 32 x odd bytes:
  // the 8 rows two lane groups touch in one LDS cycle fall on 8 disjoint bank octets (HS = 64: 160 B -> bank steps of 40).
  constexpr int VROW = HS * 2 + 32;           // bytes per key row of the V image
  extern __shared__ __attribute__((aligned(16))) char smf[];
  // two (K, V) images, used in turn: block kb + 1 is written into the other one while slower waves may still read block kb, so ONE
  // workgroup barrier per key block (image complete) is enough — every thread has finished reading image b before it passes the
  // barrier that publishes image b ^ 1, and image b is next written only after that barrier (round 6: was two barriers per block)
  constexpr int IMG = KPI * (KROW + VROW);
  const int grp = blockIdx.x, kvh = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = wave / G, head = kvh * G + (wave % G);
  const int ql = lane & 15, g = lane >> 4;
  const int row = a.group_rows[(size_t)grp * (QT * 16) + qt * 16 + ql];   // this lane's query row (-1 = padding)
  const int seq = a.group_seq[grp];
  const int nkeys = a.group_nkeys[grp];       // 1 + the largest position any row of the group attends
  const int qpos = row >= 0 ? a.row_pos[row] : -1;
  const int32_t* ptab = a.kv.page_table + (size_t)seq * a.kv.max_pages;

  // Q^T fragments: dims dc*32 + g*8 .. +8 of query row `row`, pre-scaled, split hi / lo
  u32x4 qh[DC], qlo[DC];
  {
    const float sc = 1.44269504088896340736f / sqrtf((float)HS);
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
      float f[8];
      if (row >= 0) {
        const float* qp = a.q + ((size_t)row * a.kv.n_head + head) * HS + dc * 32 + g * 8;
        const float4 t0 = *reinterpret_cast<const float4*>(qp), t1 = *reinterpret_cast<const float4*>(qp + 4);
        f[0] = t0.x; f[1] = t0.y; f[2] = t0.z; f[3] = t0.w; f[4] = t1.x; f[5] = t1.y; f[6] = t1.z; f[7] = t1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {               // hardware pair conversion (finite values: the bits of f2bf)
        unsigned hi, lo;
        split_pair(__fmul_rn(f[2 * e], sc), __fmul_rn(f[2 * e + 1], sc), hi, lo);
        qh[dc][e] = hi;
        qlo[dc][e] = lo;
      }
    }
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[DTL];
#pragma unroll
  for (int dt = 0; dt < DTL; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = (nkeys + UA2_PAGE - 1) / UA2_PAGE;
  constexpr int PPP = UA2_PAGE * HS / 8;      // 16-byte pieces per page
  constexpr int PIECES = NPG * PPP;           // ... per iteration
  constexpr int NP = (PIECES + 64 * NW - 1) / (64 * NW);      // pieces per thread
  u32x4 kk[NP], vv[NP];
  // the pages of block kb + 1 are requested right after block kb's image is complete and travel while block kb is multiplied
  auto request = [&](int kb) {                // kb = iteration: pages kb * NPG .. (a page past the last one re-reads the last: its keys are masked)
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = tid + u * 64 * NW;
      if (i < PIECES) {
        const int pg = min(kb * NPG + i / PPP, nkb - 1);
        const size_t base = (((size_t)ptab[ua2_page_slot(a.kv, pg * UA2_PAGE)] * a.kv.n_kv + kvh) * UA2_PAGE) * HS;   // elements
        kk[u] = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.kv.k_pool) + base)[i % PPP];
        vv[u] = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(a.kv.v_pool) + base)[i % PPP];
      }
    }
  };
  const int nit = (nkb + NPG - 1) / NPG;
  if (nit > 0) request(0);
  for (int kb = 0; kb < nit; ++kb) {
    char* k_lds = smf + (kb & 1) * IMG;       // [KPI keys][KROW]
    char* v_lds = k_lds + KPI * KROW;         // [KPI keys][VROW]
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int i = tid + u * 64 * NW;
      if (i < PIECES) {
        const int key = i / (HS / 8), oct = i % (HS / 8);
        *reinterpret_cast<u32x4*>(k_lds + key * KROW + oct * 16) = kk[u];
        *reinterpret_cast<u32x4*>(v_lds + key * VROW + oct * 16) = vv[u];
      }
    }
    __syncthreads();
    if (kb + 1 < nit) request(kb + 1);
    // S^T tile kt: rows = keys kb*64 + kt*16 + 4g + r, column = this lane's query
    f32x4 st[4 * NPG];
#pragma unroll
    for (int kt = 0; kt < 4 * NPG; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) {
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(k_lds + (kt * 16 + ql) * KROW + dc * 64 + g * 16));
        if constexpr (SPLIT) st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, __builtin_bit_cast(bf16x8, qlo[dc]), st[kt], 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, __builtin_bit_cast(bf16x8, qh[dc]), st[kt], 0, 0, 0);
      }
    }
    // online softmax of this lane's query over the block's 64 keys: 16 local values, then the 4 lane groups
    float mx = -INFINITY;
    // a block every query of the wave sees in full needs no mask (wave-uniform test; masking visible keys is the identity, so the
    // bits do not depend on which path a block takes)
    const bool all_visible = __all(row < 0 || qpos >= kb * KPI + KPI - 1);
    if (all_visible) {
#pragma unroll
      for (int kt = 0; kt < 4 * NPG; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
    } else {
#pragma unroll
      for (int kt = 0; kt < 4 * NPG; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kpos = kb * KPI + kt * 16 + 4 * g + r;
          if (kpos > qpos) st[kt][r] = -INFINITY;             // causal / padding mask by select: stale cache slots never leak
          mx = fmaxf(mx, st[kt][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
    u32x4 ph[2 * NPG], pl[2 * NPG];                              // P^T fragments of the 32-key chunks
#pragma unroll
    for (int kc = 0; kc < 2 * NPG; ++kc) {
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {                              // element e <-> key kc*32 + (e < 4 ? 4g + e : 16 + 4g + e - 4)
        const float sv = st[2 * kc + (e >> 2)][e & 3];
        pv[e] = (sv == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(sv - m_new);
        ps += pv[e];
      }
      // hi / lo split with the hardware pair conversion (v_cvt_pk_bf16_f32; finite values: the bits of f2bf).  The scalar
      // f2bf form was 32 calls of ~8 VALU instructions per block: the softmax took 3300 of a block's 6800 cycles
      // (cycle stamps, profiles/r3_notes.md §8)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (SPLIT) {
          unsigned hi, lo;
          split_pair(pv[2 * e], pv[2 * e + 1], hi, lo);
          ph[kc][e] = hi;
          pl[kc][e] = lo;
        } else {
          ph[kc][e] = pack_bf16x2(pv[2 * e], pv[2 * e + 1]);
        }
      }
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
    }
    // O^T tile dt (rows = dims dt*16 + 4g + r, column = this lane's query) += V^T P^T
#pragma unroll
    for (int kc = 0; kc < 2 * NPG; ++kc) {
#pragma unroll
      for (int dt = 0; dt < DTL; ++dt) {
        // this lane's piece of its group's [4 keys][16 dims] block: key kc*32 + 4g + (ql >> 2), dims dt*16 + 4 (ql & 3) .. + 3;
        // the read hands lane ql the four keys 4g .. 4g + 3 (and, 16 rows on, 16 + 4g ..) of dim dt*16 + ql
        typedef short tr4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) tr4* lds_tr4;
        const char* vr = v_lds + (size_t)(kc * 32 + 4 * g + (ql >> 2)) * VROW + (dt * 16 + 4 * (ql & 3)) * 2;
        const tr4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)vr);
        const tr4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr4)(vr + 16 * VROW));
        const uint2 v0 = __builtin_bit_cast(uint2, t0), v1 = __builtin_bit_cast(uint2, t1);                     // keys 4g..4g+3 | 16+4g..
        const bf16x8 vf = __builtin_bit_cast(bf16x8, u32x4{v0.x, v0.y, v1.x, v1.y});
        if constexpr (SPLIT) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pl[kc]), o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, ph[kc]), o[dt], 0, 0, 0);
      }
    }
  }
  if (row < 0 || l_run == 0.f) return;
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int dt = 0; dt < DTL; ++dt) {
    const int d0 = dt * 16 + 4 * g;
    const float4 out = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    if (a.y) *reinterpret_cast<float4*>(a.y + ((size_t)row * a.kv.n_head + head) * HS + d0) = out;
    if (a.y_packed) {
      const int nch = a.kv.n_head * HS / Elem<UA2_BF16>::KC;
      store_packed_operand<UA2_BF16>(a.y_packed, row, head * HS + d0, nch, out.x);
      store_packed_operand<UA2_BF16>(a.y_packed, row, head * HS + d0 + 1, nch, out.y);
      store_packed_operand<UA2_BF16>(a.y_packed, row, head * HS + d0 + 2, nch, out.z);
      store_packed_operand<UA2_BF16>(a.y_packed, row, head * HS + d0 + 3, nch, out.w);
    }
  }
}

template <int HS, int G, int QT, bool SPLIT = true, int NPG = 1>
void launch_flash_one(const ua2_attn_args& a, hipStream_t s) {
  constexpr auto kern = attn_flash_kernel<HS, G, QT, SPLIT, NPG>;
  ua2_allow_big_lds<kern>();
  const size_t smem = 2 * NPG * ((size_t)UA2_PAGE * (HS * 2 + 16) + (size_t)UA2_PAGE * (HS * 2 + 32));    // two (K, V) images
  hipLaunchKernelGGL(kern, dim3(a.n_groups, a.kv.n_kv), dim3(64 * G * QT), smem, s, a);
}

// group_q_tiles fixes QT (the host built its row lists for it): 2 for the LM's grouped-query heads, 4 for multi-head models
int launch_flash(const ua2_attn_args& a, hipStream_t s) {
  const int G = a.kv.n_head / a.kv.n_kv, hs = a.kv.head_size, qt = a.group_q_tiles;
  if (hs == 128 && G == 3 && qt == 2) launch_flash_one<128, 3, 2>(a, s);
  else if (hs == 128 && G == 1 && qt == 4) launch_flash_one<128, 1, 4>(a, s);
  else if (hs == 64 && G == 1 && qt == 4) launch_flash_one<64, 1, 4>(a, s);
  else if (hs == 64 && G == 1 && qt == 8 && (a.flags & UA2_ATTN_BF16_QP)) {
    static const bool one_page = getenv("UA2_ATTN_ONE_PAGE") != nullptr;      // A/B hook: 64 keys per iteration (the round-5 form)
    // two pages per iteration where the launch has at most one workgroup per CU (one 20-s window: 192 workgroups, DiT step 5.05 -> 5.02 ms);
    // with more, the smaller images (two workgroups per CU at 39 KiB) stay
    static const bool two_pages = getenv("UA2_ATTN_TWO_PAGES") != nullptr;    // A/B hook: 128 keys per iteration whatever the grid
    if (one_page || (!two_pages && (int64_t)a.n_groups * a.kv.n_kv > 256)) launch_flash_one<64, 1, 8, false>(a, s);
    else launch_flash_one<64, 1, 8, false, 2>(a, s);
  }
  else if (hs == 64 && G == 1 && qt == 8) launch_flash_one<64, 1, 8>(a, s);
  else if (hs == 64 && G == 2 && qt == 2) launch_flash_one<64, 2, 2>(a, s);
  else if (hs == 64 && G == 4 && qt == 2) launch_flash_one<64, 4, 2>(a, s);
  else if (hs == 128 && G == 2 && qt == 2) launch_flash_one<128, 2, 2>(a, s);
  else if (hs == 32 && G == 2 && qt == 2) launch_flash_one<32, 2, 2>(a, s);
  else if (hs == 32 && G == 1 && qt == 4) launch_flash_one<32, 1, 4>(a, s);
  else {
    ua2_set_error("ua2_attn: no grouped (flash) kernel for head_size %d, group %d, q tiles %d", hs, G, qt);
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
  UA2_CHECK(a.kv.ring_pages == 0 || (a.window > 0 && a.kv.ring_pages <= a.kv.max_pages && a.window <= (a.kv.ring_pages - 1) * UA2_PAGE + 1 &&
                                     (a.kv.ring_pages & (a.kv.ring_pages - 1)) == 0),
            "ua2_attn: a ring cache (ring_pages=%d) needs a power-of-two page count and 0 < window <= (ring_pages - 1) * %d + 1", a.kv.ring_pages, UA2_PAGE);
  UA2_CHECK(!a.y_packed || (a.kv.n_head * a.kv.head_size) % (a.dtype == UA2_BF16 ? 32 : 16) == 0, "ua2_attn: y_packed needs n_head*head_size %% chunk == 0");
  if (a.group_rows && a.n_groups > 0 && a.dtype == UA2_BF16 && a.window <= 0) {   // many rows per sequence: MFMA flash form
    UA2_CHECK(a.group_seq && a.group_nkeys && a.group_q_tiles > 0, "ua2_attn: group_seq / group_nkeys / group_q_tiles missing");
    return launch_flash(a, s);
  }
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
