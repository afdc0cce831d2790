// ua2_linear: Y[M,N] = epilogue( prologue(X)[M,K] * W[N,K]^T ) — weight packing, argument validation and dispatch.
//
// Replaces the nn.Linear call sites of the reference's decode frame together with the op right before and
// right after each of them (include/ua2hip.h lists the file:line of every
// fused piece).  The kernels live in ua2_gemv.hip (decode regime: one row tile, activations staged in LDS,
// the weight streamed once as non-temporal 1 KiB fragment bursts) and ua2_gemm.hip (many rows: packed
// operand, skinny or 128-row tiled MFMA GEMM); all of them give the same bits per row.
//   * the weight is pre-tiled here into MFMA B-fragment order (ua2_pack_linear): every wave load is one
//     contiguous 1 KiB burst (64 lanes x 16 B);
//   * M rows ride in the 16-row A operand of mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulate) or
//     mfma_f32_16x16x4_f32 (exact fp32).
#include "ua2_common.h"
#include "ua2_linear_common.h"

namespace {

// ---- packing ------------------------------------------------------------------------------

template <int SRC, int DST>
__global__ void pack_kernel(const void* __restrict__ src, void* __restrict__ out, int transposed, int64_t N, int64_t K,
                            int64_t total, int rope_hs) {
  constexpr int KC = Elem<DST>::KC, EPL = Elem<DST>::EPL;
  const int64_t nchunks = (K + KC - 1) / KC;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx % EPL);
    const int lane = (int)((idx / EPL) % 64);
    const int64_t chunk = (idx / (EPL * 64)) % nchunks;
    const int64_t tile = idx / (EPL * 64 * nchunks);
    int64_t n = tile * 16 + (lane & 15);
    if (rope_hs > 0) {  // packed column -> source row: [8r, 8r+8) then [hs/2+8r, hs/2+8r+8) per tile r of a head
      const int64_t h = n / rope_hs, within = n - h * rope_hs;
      const int64_t r = within / 16, c = within % 16;
      n = h * rope_hs + (c < 8 ? r * 8 + c : rope_hs / 2 + r * 8 + (c - 8));
    }
    const int64_t k = chunk * KC + (lane >> 4) * EPL + e;
    float v = 0.f;
    if (n < N && k < K) v = load_elem<SRC>(src, transposed ? (size_t)(k * N + n) : (size_t)(n * K + k));
    store_elem<DST>(out, (size_t)idx, v);
  }
}

}  // namespace

static int g_force_general = 0;

int ua2_linear_launch(const ua2_linear_args& a, hipStream_t s) {
  UA2_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "ua2_linear: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  UA2_CHECK(a.w0 != nullptr, "ua2_linear: w0 is NULL");
  const int epl = a.dtype == UA2_BF16 ? 8 : 4;
  UA2_CHECK(a.K % epl == 0, "ua2_linear: K=%d must be a multiple of %d", a.K, epl);
  if (a.prologue == UA2_PRO_LOCAL_ATTN) {
    const int kc = a.dtype == UA2_BF16 ? 32 : 16;
    UA2_CHECK(a.M == 1 && a.epilogue == UA2_EPI_RESIDUAL, "ua2_linear: LOCAL_ATTN is the M == 1 O-projection only (use ua2_attn_local + CAST otherwise)");
    UA2_CHECK(a.x && a.row_pos && a.kv.k_pool && a.kv.v_pool && a.kv.page_table && a.kv.n_kv > 0 && a.kv.n_head % a.kv.n_kv == 0 &&
                  (a.kv.head_size == 32 || a.kv.head_size == 64 || a.kv.head_size == 128) &&
                  a.kv.n_head % (128 / a.kv.head_size) == 0 && a.K == a.kv.n_head * a.kv.head_size && a.K % kc == 0,
              "ua2_linear: bad LOCAL_ATTN arguments");
    UA2_CHECK(a.resid != nullptr && a.y != nullptr, "ua2_linear: RESIDUAL needs resid, y");
    const int rc = ua2_gemv_try_launch(a, s);
    UA2_CHECK(rc <= 0, "ua2_linear: LOCAL_ATTN problem outside the decode kernel's range");
    return rc;
  }
  if (a.y_norm_w) {   // producer half of the scaled-norm hand-over
    UA2_CHECK(a.dtype == UA2_BF16 && (a.epilogue == UA2_EPI_RESIDUAL || a.epilogue == UA2_EPI_STORE) && a.N % 32 == 0 && a.y_ssq &&
                  (a.y_h || a.y_packed) && (!a.y_h || a.ldh % 8 == 0),
              "ua2_linear: y_norm_w hand-over needs UA2_BF16, a RESIDUAL / STORE epilogue, N %% 32 == 0, y_ssq and y_h (ldh %% 8 == 0) or y_packed");
  }
  if (a.y_ln_w)     // LayerNorm hand-over: the order-free kernel's form only (ua2_linear_order_free_accepts tells a caller beforehand)
    UA2_CHECK(a.dtype == UA2_BF16 && a.sum_order == UA2_SUM_ORDER_FREE && a.epilogue == UA2_EPI_RESIDUAL && a.y_packed && !a.y_norm_w && a.N % 32 == 0,
              "ua2_linear: y_ln_w needs UA2_BF16, UA2_SUM_ORDER_FREE, a RESIDUAL epilogue, y_packed and N %% 32 == 0");
  if (a.prologue == UA2_PRO_SCALED) {
    UA2_CHECK(a.dtype == UA2_BF16 && a.K % 32 == 0 && a.K <= 4096 && a.x_ssq && (a.x_h || a.x_packed) && (!a.x_h || a.ldh % 8 == 0),
              "ua2_linear: UA2_PRO_SCALED needs UA2_BF16, K %% 32 == 0, K <= 4096, x_ssq and x_h (ldh %% 8 == 0) or x_packed");
    UA2_CHECK(a.epilogue == UA2_EPI_QKV_ROPE || a.epilogue == UA2_EPI_SWIGLU || a.epilogue == UA2_EPI_STORE,
              "ua2_linear: UA2_PRO_SCALED serves the QKV_ROPE, SWIGLU and STORE epilogues");
    // The tiled kernel's staged no-rotation head-size-64 q|k|v epilogue (the DiT's) writes the un-scaled sums: no caller
    // pairs it with the scaled hand-over, and the ABI refuses the pair instead of returning un-normalised q / k / v.
    UA2_CHECK(a.epilogue != UA2_EPI_QKV_ROPE || a.rope_mode != UA2_ROPE_NONE,
              "ua2_linear: UA2_PRO_SCALED with a QKV_ROPE epilogue needs a rotation mode (UA2_ROPE_NONE is served by the NORM / CAST prologues)");
  }
  if (a.x_packed) {   // operand handed over in fragment order by its producer: only the many-row kernels read it
    UA2_CHECK((a.prologue == UA2_PRO_CAST || a.prologue == UA2_PRO_SCALED) && g_force_general != 2,
              "ua2_linear: x_packed needs PRO_CAST / PRO_SCALED and the many-row kernels");
    const int rc = ua2_gemm_try_launch(a, s, g_force_general >= 4 ? g_force_general : 3);
    UA2_CHECK(rc <= 0, "ua2_linear: x_packed launch not applicable");
    return rc;
  }
  if (a.prologue == UA2_PRO_SCALED) {   // row-major hand-over: one row tile, the decode kernel
    UA2_CHECK(a.M <= ua2_gemv_rows_per_tile(a.dtype, a.K), "ua2_linear: x_h serves launches of one row tile (M=%d): hand over x_packed", a.M);
    const int rc = ua2_gemv_try_launch(a, s);
    UA2_CHECK(rc <= 0, "ua2_linear: UA2_PRO_SCALED launch outside the decode kernel's range");
    return rc;
  }
  UA2_CHECK(a.prologue == UA2_PRO_CAST || a.prologue == UA2_PRO_NORM, "ua2_linear: bad prologue %d", a.prologue);
  UA2_CHECK(a.x != nullptr && a.ldx % 4 == 0, "ua2_linear: x NULL or ldx %% 4 != 0");
  if (a.prologue == UA2_PRO_NORM)
    UA2_CHECK(a.norm_w != nullptr && (a.norm_kind != UA2_NORM_LAYERNORM || a.norm_b != nullptr) && a.norm_kind >= 0 && a.norm_kind <= 2,
              "ua2_linear: norm_w / norm_b / norm_kind invalid");
  if (a.epilogue == UA2_EPI_GELU) {
    UA2_CHECK(a.y != nullptr || a.y_packed != nullptr, "ua2_linear: GELU needs y or y_packed");
    UA2_CHECK(!a.y_packed || a.N % (a.dtype == UA2_BF16 ? 32 : 16) == 0, "ua2_linear: y_packed needs N %% chunk == 0");
  }
  UA2_CHECK(!a.bias || a.epilogue != UA2_EPI_QKV_ROPE || a.rope_mode != UA2_ROPE_HALF_SPLIT,
            "ua2_linear: bias with the half-split QKV layout is not supported (columns are permuted at pack time)");
  UA2_CHECK(a.act_kind >= 0 && a.act_kind <= 2, "ua2_linear: bad act_kind %d", a.act_kind);
  if (a.epilogue == UA2_EPI_SWIGLU) {
    UA2_CHECK(a.w1 != nullptr && (a.y != nullptr || a.y_packed != nullptr), "ua2_linear: SWIGLU needs w1 and y or y_packed");
    UA2_CHECK(!a.y_packed || a.N % (a.dtype == UA2_BF16 ? 32 : 16) == 0, "ua2_linear: y_packed needs N %% chunk == 0");
  } else if (a.epilogue != UA2_EPI_GELU) {
    UA2_CHECK(!a.y_packed || a.y_norm_w || a.y_ln_w, "ua2_linear: y_packed is a SWIGLU / GELU output (or, with y_norm_w / y_ln_w, a RESIDUAL / STORE hand-over)");
  }
  if (a.epilogue == UA2_EPI_RESIDUAL) UA2_CHECK(a.resid != nullptr && a.y != nullptr, "ua2_linear: RESIDUAL needs resid, y");
  if (a.epilogue == UA2_EPI_STORE) UA2_CHECK(a.y != nullptr || a.part_max != nullptr, "ua2_linear: STORE needs y or part_max");
  if (a.epilogue == UA2_EPI_QKV_ROPE) {
    UA2_CHECK(a.kv.head_size % (a.rope_mode == UA2_ROPE_HALF_SPLIT ? 32 : 16) == 0 &&
                  a.N == (a.kv.n_head + 2 * a.kv.n_kv) * a.kv.head_size && a.rope_mode >= 0 && a.rope_mode <= 2,
              "ua2_linear: QKV_ROPE needs head_size %% 32 == 0 (16 when not half-split) and N == (n_head+2*n_kv)*head_size");
    UA2_CHECK(a.row_pos && (a.rope_mode == UA2_ROPE_NONE || (a.rope_cos && a.rope_sin)) && a.q_out && a.kv.k_pool && a.kv.v_pool &&
                  a.kv.page_table,
              "ua2_linear: QKV_ROPE pointer arguments missing");
    UA2_CHECK(a.kv.ring_pages == 0 || (a.kv.ring_pages & (a.kv.ring_pages - 1)) == 0, "ua2_linear: ring_pages=%d must be a power of two", a.kv.ring_pages);
  }
  if (a.dtype != UA2_BF16 && a.dtype != UA2_F32) {
    ua2_set_error("ua2_linear: bad dtype %d", a.dtype);
    return -1;
  }
  if (g_force_general != 2) {
    const int rc = ua2_gemm_try_launch(a, s, g_force_general >= 3 ? g_force_general : 0);  // many rows: packed operand, 128-row tiles
    if (rc <= 0) return rc;
  }
  const int rc = ua2_gemv_try_launch(a, s);  // decode regime: LDS-staged activations, all loads up front
  UA2_CHECK(rc <= 0, "ua2_linear: prologue %d / epilogue %d / K=%d is outside the built kernels (a row must fit the LDS operand tile)",
            a.prologue, a.epilogue, a.K);
  return rc;
}

extern "C" int ua2_debug_force_general_linear(int on) {
  const int old = g_force_general;
  g_force_general = (on == 1) ? 0 : on;   // mode 1 (the round-1 general-M kernel) no longer exists
  return old;
}

extern "C" int ua2_linear(const ua2_linear_args* a, void* stream) {
  UA2_CHECK(a != nullptr, "ua2_linear: NULL args");
  return ua2_linear_launch(*a, (hipStream_t)stream);
}

extern "C" int ua2_linear_chain_timed(const ua2_linear_args* args, int32_t n, int32_t iters, void* stream,
                                      float* ms_out) {
  UA2_CHECK(args && n > 0 && iters > 0 && ms_out, "ua2_linear_chain_timed: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  UA2_HIP(hipEventCreate(&e0));
  UA2_HIP(hipEventCreate(&e1));
  UA2_HIP(hipEventRecord(e0, s));
  for (int it = 0; it < iters; ++it)
    for (int i = 0; i < n; ++i)
      if (int rc = ua2_linear_launch(args[i], s)) return rc;
  UA2_HIP(hipEventRecord(e1, s));
  UA2_HIP(hipEventSynchronize(e1));
  UA2_HIP(hipEventElapsedTime(ms_out, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

extern "C" size_t ua2_packed_elems(int dtype, int64_t N, int64_t K) {
  const int kc = dtype == UA2_BF16 ? 32 : 16, epl = dtype == UA2_BF16 ? 8 : 4;
  return (size_t)((N + 15) / 16) * (size_t)((K + kc - 1) / kc) * 64 * epl;
}

extern "C" int ua2_pack_linear(const void* src, int src_dtype, int transposed, int64_t N, int64_t K, void* out,
                               int dtype, int rope_head_size, void* stream) {
  UA2_CHECK(src && out && N > 0 && K > 0, "ua2_pack_linear: bad arguments");
  UA2_CHECK(rope_head_size == 0 || (rope_head_size % 32 == 0 && N % rope_head_size == 0),
            "ua2_pack_linear: rope_head_size=%d must divide N and be a multiple of 32", rope_head_size);
  const int64_t total = (int64_t)ua2_packed_elems(dtype, N, K);
  const int threads = 256;
  const int blocks = (int)((total + threads - 1) / threads < 65535 * 16 ? (total + threads - 1) / threads : 65535 * 16);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == UA2_F32 && dtype == UA2_F32)
    hipLaunchKernelGGL((pack_kernel<UA2_F32, UA2_F32>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else if (src_dtype == UA2_F32 && dtype == UA2_BF16)
    hipLaunchKernelGGL((pack_kernel<UA2_F32, UA2_BF16>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else if (src_dtype == UA2_BF16 && dtype == UA2_BF16)
    hipLaunchKernelGGL((pack_kernel<UA2_BF16, UA2_BF16>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else if (src_dtype == UA2_BF16 && dtype == UA2_F32)
    hipLaunchKernelGGL((pack_kernel<UA2_BF16, UA2_F32>), dim3(blocks), dim3(threads), 0, s, src, out, transposed, N, K, total, rope_head_size);
  else {
    ua2_set_error("ua2_pack_linear: bad dtypes %d -> %d", src_dtype, dtype);
    return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}
