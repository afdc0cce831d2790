// Internal helpers shared by the gfx950 kernels of libua2hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/ua2hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define UA2_WAVE 64

void ua2_set_error(const char* fmt, ...);

#define UA2_CHECK(cond, ...)                 \
  do {                                       \
    if (!(cond)) {                           \
      ua2_set_error(__VA_ARGS__);            \
      return -1;                             \
    }                                        \
  } while (0)

#define UA2_HIP(call)                                                          \
  do {                                                                         \
    hipError_t _e = (call);                                                    \
    if (_e != hipSuccess) {                                                    \
      ua2_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                               \
    }                                                                          \
  } while (0)

#define UA2_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess) {                                                    \
      ua2_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return -3;                                                               \
    }                                                                          \
  } while (0)

// Per-dtype tiling constants.  One wave-level "chunk" = one 16-byte load per lane of the
// packed weight = KC values of K for 16 output columns.
template <int DT> struct Elem;
template <> struct Elem<UA2_BF16> {
  static constexpr int KC = 32;   // K per chunk (one mfma_f32_16x16x32_bf16)
  static constexpr int EPL = 8;   // elements per lane per chunk
  static constexpr int BYTES = 2;
};
template <> struct Elem<UA2_F32> {
  static constexpr int KC = 16;   // four mfma_f32_16x16x4_f32
  static constexpr int EPL = 4;
  static constexpr int BYTES = 4;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() / s_barrier through the compiler costs more than it looks
// on gfx9-family targets: the waitcnt pass puts a full `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every s_barrier it sees
// (no automatic wait-before-barrier in hardware), which drains every global load a kernel meant to keep in flight across
// the barrier — register prefetch of the next tile, weight refills, the lot.  Spelled as inline asm the barrier is opaque
// to that pass; the LDS writes the barrier publishes are waited for explicitly.  Use only where the data exchanged
// through the barrier lives in LDS (global-memory hand-offs between waves still need __syncthreads()).
__device__ __forceinline__ void ua2_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fp32 -> bf16 bits, round to nearest even (same as torch's .to(torch.bfloat16)).
__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
// (lo half = bf16(a), hi half = bf16(b)), round to nearest even: one v_cvt_pk_bf16_f32 (finite values: same bits as f2bf)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const f32x2_hw v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_hw));
}
// hi/lo split of a pair: hi = RNE(x), lo = RNE(x - hi)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2(x0, x1);
  lo = pack_bf16x2(__fsub_rn(x0, __uint_as_float(hi << 16)), __fsub_rn(x1, __uint_as_float(hi & 0xffff0000u)));
}

template <int DT> __device__ __forceinline__ float load_elem(const void* p, size_t i);
template <> __device__ __forceinline__ float load_elem<UA2_F32>(const void* p, size_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float load_elem<UA2_BF16>(const void* p, size_t i) {
  return bf2f(((const unsigned short*)p)[i]);
}
template <int DT> __device__ __forceinline__ void store_elem(void* p, size_t i, float v);
template <> __device__ __forceinline__ void store_elem<UA2_F32>(void* p, size_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void store_elem<UA2_BF16>(void* p, size_t i, float v) {
  ((unsigned short*)p)[i] = f2bf(v);
}

// element (row m, column k) of an [M, K] operand in MFMA fragment order [M/16][K/KC][64 lanes][16 B]
template <int DT>
__device__ __forceinline__ void store_packed_operand(void* base, int m, int k, int nchunks, float v) {
  constexpr int KC = Elem<DT>::KC, EPL = Elem<DT>::EPL;
  const int c = k / KC, r = k - c * KC, g = r / EPL, e = r - g * EPL;
  const size_t elem = (((size_t)(m >> 4) * nchunks + c) * 64 + g * 16 + (m & 15)) * EPL + e;
  store_elem<DT>(base, elem, v);
}

// four consecutive elements (row m, columns k .. k + 3, k % 4 == 0) of the same layout: one 8-byte (bf16) / 16-byte (fp32) store
template <int DT>
__device__ __forceinline__ void store_packed4(void* base, int m, int k, int nchunks, const float4& v) {
  constexpr int KC = Elem<DT>::KC, EPL = Elem<DT>::EPL;
  const int c = k / KC, r = k - c * KC, g = r / EPL, e = r - g * EPL;
  const size_t elem = (((size_t)(m >> 4) * nchunks + c) * 64 + g * 16 + (m & 15)) * EPL + e;
  if constexpr (DT == UA2_BF16) {
    uint2 p;
    p.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    p.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + elem) = p;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem) = v;
  }
}
// four consecutive elements of a row-major array of the launch dtype
template <int DT>
__device__ __forceinline__ void store_row4(void* base, size_t i, const float4& v) {
  if constexpr (DT == UA2_BF16) {
    uint2 p;
    p.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    p.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + i) = p;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + i) = v;
  }
}

// Page-table column of position `pos`: linear caches index by pos / 64; ring caches (kv.ring_pages > 0: the streaming
// transformers of the Moshi family, llm_modules/transformer.py:211-278) wrap over ring_pages pages.
__device__ __forceinline__ int ua2_page_slot(const ua2_kv_geom& kv, int pos) {
  const int lp = pos / UA2_PAGE;
  return kv.ring_pages > 0 ? (lp & (kv.ring_pages - 1)) : lp;   // ring_pages is a power of two (checked by the launchers): no integer division per key
}

static inline int ua2_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised.  The
// attribute is per device, so it is set once per (kernel instantiation, device ordinal); safe to race (the call is
// idempotent, the flag only saves the repeat).
template <auto Kern>
inline void ua2_allow_big_lds() {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.fetch_or(bit, std::memory_order_release);
}

// ---- test hooks (ua2hip.h ABI v9): launch counters per kernel family, and UA2_* environment variables read once -------------
enum { UA2_CNT_GEMM2 = 0, UA2_CNT_GEMM = 1, UA2_CNT_SKINNY2 = 2, UA2_CNT_GEMV = 3, UA2_CNT_RSPLIT = 4, UA2_CNT_N = 5 };
extern std::atomic<int64_t> g_ua2_launches[UA2_CNT_N];
extern std::atomic<int> g_ua2_env_gen;              // bumped by ua2_debug_refresh_env
inline void ua2_count_launch(int family) { g_ua2_launches[family].fetch_add(1, std::memory_order_relaxed); }
// An integer knob read from the environment at first use and again after ua2_debug_refresh_env (launchers used to call getenv
// several times per launch).  `unset` is returned while the variable is absent or empty.  Benign if two threads race on the first read.
struct Ua2EnvInt {
  const char* name; int unset; int gen = -1; int val = 0; bool present = false;
  constexpr Ua2EnvInt(const char* n, int u) : name(n), unset(u) {}
  void load() {
    const int g = g_ua2_env_gen.load(std::memory_order_acquire);
    if (g == gen) return;
    const char* e = getenv(name);
    present = e != nullptr;
    val = (e && *e) ? atoi(e) : unset;
    gen = g;
  }
  int get() { load(); return val; }
  bool set() { load(); return present; }
};

// internal launchers used by both the op-level ABI and the frame executor
int ua2_linear_launch(const ua2_linear_args& a, hipStream_t s);
int ua2_attn_launch(const ua2_attn_args& a, hipStream_t s);
int ua2_attn_local_launch(const ua2_attn_args& a, hipStream_t s);
int ua2_gemv_rows_per_tile(int dtype, int K);   // rows one decode-kernel workgroup holds in LDS (ua2_gemv.hip)
int ua2_gemv_rows_preferred(int dtype, int K);  // rows up to which the launchers prefer the decode kernel (<= rows_per_tile; a cost choice)
extern "C" int ua2_sample_topk(int dtype, int32_t M, const float* logits, int32_t ld, int32_t V, int32_t topk, float temperature,
                               const int32_t* forbid, uint64_t seed, const int32_t* counter, int32_t stream_id,
                               int32_t* out_tokens, int32_t out_ld, int32_t out_col, const void* emb, int32_t emb_row_offset,
                               int32_t C, float* next_h, int32_t row_key_shift, void* stream);
extern "C" int ua2_cfg_mix(float* logits, int32_t ld, int32_t V, float scale, const int32_t* forbid, float* part_max,
                           int32_t* part_idx, int32_t pairs, void* stream);
// arg-max over the STORE epilogue's partials fused with a gather from the executor's projected-embedding table (ua2_misc.hip)
// ... and, optionally, layer 0's q | k | v of the step (also functions of the id and the step's position): q -> q_out [M, qn], k / v -> the caches
struct ua2_qkv_gather {
  const float* tab_q;      // [rows, qn]
  const void* tab_k;       // [rows, n_kv * head_size] of the plan dtype
  const void* tab_v;
  float* q_out;
  int32_t qn, esz, pos;    // esz: bytes per cache element; pos: position of the step the rows are written for
  ua2_kv_geom kv;          // destination caches (row m = sequence m)
};
int ua2_argmax_gather(int32_t M, int32_t n_part, const float* part_max, const int32_t* part_idx, int32_t* out_tokens, int32_t out_ld,
                      int32_t out_col, const float* tab_y, const void* tab_h, const float* tab_ssq, int64_t row_off, int32_t Cd, float* next_x,
                      const ua2_handover* ho, const ua2_qkv_gather* qg, hipStream_t s);
int ua2_kv_rows_extract(const void* k_pool, const void* v_pool, const int32_t* pos, int n, int n_kv, int hs, int esz, void* out_k, void* out_v, hipStream_t s);
// decode-regime specialisation; returns 1 when the problem is outside its regime
int ua2_gemv_try_launch(const ua2_linear_args& a, hipStream_t s);
// riders (ua2_gemv.hip gemv_rider_kernel): column tiles [tile0, tile1) of the one-row-tile GEMV `r` on the idle CUs of host launch `a`
bool ua2_gemv_rider_ok(const ua2_linear_args& a, const ua2_linear_args& r);
int ua2_gemv_launch_with_rider(const ua2_linear_args& a, const ua2_linear_args& r, int tile0, int tile1, hipStream_t s);
