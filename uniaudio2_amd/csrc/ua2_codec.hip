// Small row-wise kernels of the codec's neural stages (ReasoningCodec_film): the glue between the GEMMs of the
// flow-matching DiT (models/transformer_1d_flow.py, models/attention.py) and of the AudioThinking encoder
// (modules/transformer.py:645-783, models/AudioDiffusion1D.py:372-390,428-456).  All of them move a few MB at most;
// each replaces a chain of tiny PyTorch launches (chunk / broadcast-multiply / add / tanh / interpolate / masked blend).
#include "ua2_common.h"

namespace {

// out[i] = alpha * a[i % na] * (b ? b[i % nb] : 1) + (c ? c[i % nc] : 0) + beta
// One kernel for every broadcast multiply-add of the DiT / Euler solver:
//   adaLN modulation vectors  1 + (table + t_emb)                 attention.py:308-311
//   gated residual            gate * branch + hidden               attention.py:345-349,401-405
//   ProjectLayer scale        x * kernel_size^-0.5                 transformer_1d_flow.py:31
//   guidance                  u + s * (c - u), Euler  x + dt * d   AudioDiffusion1D.py:116-123
//   in-context blend          (1 - (1 - sigma) t) * noise + t * x  AudioDiffusion1D.py:104
__global__ void ew_fma_kernel(float* __restrict__ out, int64_t n, const float* __restrict__ a, int64_t na,
                              const float* __restrict__ b, int64_t nb, const float* __restrict__ c, int64_t nc, float alpha,
                              float beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = __fmul_rn(alpha, a[i % na]);
    if (b) v = __fmul_rn(v, b[i % nb]);
    if (c) v = __fadd_rn(v, c[i % nc]);
    out[i] = __fadd_rn(v, beta);
  }
}

__global__ void ew_act_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    float r;
    switch (act) {
      case UA2_EW_SILU: r = v / (1.0f + expf(-v)); break;
      case UA2_EW_SIGMOID: r = 1.0f / (1.0f + expf(-v)); break;
      case UA2_EW_TANH: r = tanhf(v); break;
      default: r = v;
    }
    out[i] = r;
  }
}

// out[r, :] = in[idx[r], :]   (idx < 0: zeros).  F.interpolate(mode="nearest") along time (AudioDiffusion1D.py:450,512,590)
// with the source indices computed on the host by torch's own rule; also the cls-token interleave / extraction of
// set_masking / extract_mask_positions (:458-486).
__global__ void gather_rows_kernel(float* __restrict__ out, const float* __restrict__ in, const int32_t* __restrict__ idx, int C) {
  const int64_t r = blockIdx.x;
  const int32_t s = idx[r];
  for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) v = *reinterpret_cast<const float4*>(in + (size_t)s * C + c);
    *reinterpret_cast<float4*>(out + (size_t)r * C + c) = v;
  }
}

// time_film (AudioDiffusion1D.py:428-438): params [R, 2C] = (delta_gamma | beta);
//   gamma = 1 + g * tanh(delta_gamma); rows of a masked batch element use gamma = 1, beta = 0;  out = gamma * x + beta
__global__ void time_film_kernel(float* __restrict__ out, const float* __restrict__ params, const float* __restrict__ x,
                                 const uint8_t* __restrict__ batch_mask, int rows_per_batch, int C, float g) {
  const int64_t r = blockIdx.x;
  const bool masked = batch_mask && batch_mask[r / rows_per_batch];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float gamma = 1.0f + g * tanhf(params[(size_t)r * 2 * C + c]);
    float beta = params[(size_t)r * 2 * C + C + c];
    if (masked) { gamma = 1.0f; beta = 0.0f; }
    out[(size_t)r * C + c] = gamma * x[(size_t)r * C + c] + beta;
  }
}

// LayerNorm over the last axis, optional affine (w, b may be NULL): F.layer_norm semantics, biased variance.
// One workgroup per row; two-pass (mean, then centred sum of squares) in fp32.
__global__ __launch_bounds__(256) void layernorm_rows_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, int C, float eps) {
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const float* xr = x + (size_t)r * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int c = tid; c < C; c += 256) s += xr[c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (((red[0] + red[1]) + red[2]) + red[3]) / (float)C;
  __syncthreads();
  float q = 0.f;
  for (int c = tid; c < C; c += 256) { const float d = xr[c] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)C + eps);
  for (int c = tid; c < C; c += 256) {
    float v = (xr[c] - mean) * rstd;
    if (w) v *= w[c];
    if (b) v += b[c];
    out[(size_t)r * C + c] = v;
  }
}

// q/k LayerNorm over the head dimension + partial rotary + K/V append to the paged cache, for the x-transformers
// style Attention of the AudioThinking encoder (modules/transformer.py:452-485: `q_norm`, `k_norm` = nn.LayerNorm
// (dim_heads), then apply_rotary_pos_emb on the first rot_dim dims with rotate_half over those dims, :146-170).
// qkv [R, 3 * n_head * hs] = (q | k | v), heads contiguous inside each third (:447-448).  One workgroup per
// (row, head), one thread per dim.  norm weights NULL: no q/k norm.  rot_dim 0: no rotary.
template <int DT>
__global__ void qknorm_rope_kv_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ row_pos,
                                      const int32_t* __restrict__ row_seq, const float* __restrict__ qw, const float* __restrict__ qb,
                                      const float* __restrict__ kw, const float* __restrict__ kb, float eps,
                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t, int rot_dim,
                                      float* __restrict__ q_out, ua2_kv_geom kv) {
  extern __shared__ float sm[];          // [2][hs] values, then [2][4] partials
  const int r = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
  const int hs = kv.head_size, dim = kv.n_head * hs;
  const int nw = (hs + 63) >> 6, lane = d & 63, wave = d >> 6;
  float* qs = sm;
  float* ks = sm + hs;
  float* part = sm + 2 * hs;             // [4][nw]: sum q, sum k, then centred squares
  const float* row = qkv + (size_t)r * 3 * dim;
  float q = row[h * hs + d], k = row[dim + h * hs + d];
  const float v = row[2 * dim + h * hs + d];
  if (qw) {
    float sq = q, sk = k;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { sq += __shfl_xor(sq, o); sk += __shfl_xor(sk, o); }
    if (lane == 0) { part[wave] = sq; part[4 + wave] = sk; }
    __syncthreads();
    float mq = 0.f, mk = 0.f;
    for (int w = 0; w < nw; ++w) { mq += part[w]; mk += part[4 + w]; }
    mq /= (float)hs; mk /= (float)hs;
    const float dq = q - mq, dk = k - mk;
    float vq = dq * dq, vk = dk * dk;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { vq += __shfl_xor(vq, o); vk += __shfl_xor(vk, o); }
    __syncthreads();
    if (lane == 0) { part[wave] = vq; part[4 + wave] = vk; }
    __syncthreads();
    float tq = 0.f, tk = 0.f;
    for (int w = 0; w < nw; ++w) { tq += part[w]; tk += part[4 + w]; }
    q = dq * (1.0f / sqrtf(tq / (float)hs + eps)) * qw[d] + qb[d];
    k = dk * (1.0f / sqrtf(tk / (float)hs + eps)) * kw[d] + kb[d];
  }
  const int pos = row_pos[r];
  if (rot_dim > 0) {
    qs[d] = q; ks[d] = k;
    __syncthreads();
    if (d < rot_dim) {
      const int half = rot_dim / 2;
      const int f = d % half;                                   // freqs = cat(freqs, freqs)
      const float c = cos_t[(size_t)pos * half + f], s = sin_t[(size_t)pos * half + f];
      const float qp = d < half ? -qs[d + half] : qs[d - half];  // rotate_half: cat(-x2, x1)
      const float kp = d < half ? -ks[d + half] : ks[d - half];
      q = q * c + qp * s;
      k = k * c + kp * s;
    }
  }
  q_out[(size_t)r * dim + h * hs + d] = q;
  const int seq = row_seq ? row_seq[r] : r;
  const int page = kv.page_table[(size_t)seq * kv.max_pages + ua2_page_slot(kv, pos)];
  const size_t base = (((size_t)page * kv.n_kv + h) * UA2_PAGE + (pos % UA2_PAGE)) * hs + d;
  store_elem<DT>(kv.k_pool, base, k);
  store_elem<DT>(kv.v_pool, base, v);
}

int blocks_for(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 4096); }

}  // namespace

extern "C" int ua2_ew_fma(float* out, int64_t n, const float* a, int64_t na, const float* b, int64_t nb, const float* c,
                          int64_t nc, float alpha, float beta, void* stream) {
  UA2_CHECK(out && a && n > 0 && na > 0 && (!b || nb > 0) && (!c || nc > 0), "ua2_ew_fma: bad arguments");
  hipLaunchKernelGGL(ew_fma_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, n, a, na, b, nb, c, nc, alpha, beta);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_ew_act(float* out, const float* x, int64_t n, int32_t act, void* stream) {
  UA2_CHECK(out && x && n > 0 && act >= 0 && act <= UA2_EW_TANH, "ua2_ew_act: bad arguments");
  hipLaunchKernelGGL(ew_act_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, x, n, act);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_gather_rows(float* out, const float* in, const int32_t* idx, int64_t R, int32_t C, void* stream) {
  UA2_CHECK(out && in && idx && R > 0 && C > 0 && C % 4 == 0, "ua2_gather_rows: bad arguments (C %% 4 == 0)");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, out, in, idx, C);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_time_film(float* out, const float* params, const float* x, const uint8_t* batch_mask, int64_t R,
                             int32_t rows_per_batch, int32_t C, float gamma_scale, void* stream) {
  UA2_CHECK(out && params && x && R > 0 && C > 0 && rows_per_batch > 0, "ua2_time_film: bad arguments");
  hipLaunchKernelGGL(time_film_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, out, params, x, batch_mask, rows_per_batch, C,
                     gamma_scale);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_layernorm_rows(float* out, const float* x, const float* w, const float* b, int64_t R, int32_t C, float eps,
                                  void* stream) {
  UA2_CHECK(out && x && R > 0 && C > 0, "ua2_layernorm_rows: bad arguments");
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, out, x, w, b, C, eps);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_qknorm_rope_kv(int dtype, const float* qkv, int64_t R, const int32_t* row_pos, const int32_t* row_seq,
                                  const float* q_norm_w, const float* q_norm_b, const float* k_norm_w, const float* k_norm_b,
                                  float eps, const float* cos_t, const float* sin_t, int32_t rot_dim, float* q_out,
                                  const ua2_kv_geom* kv, void* stream) {
  UA2_CHECK(qkv && row_pos && q_out && kv && kv->k_pool && kv->v_pool && kv->page_table && R > 0, "ua2_qknorm_rope_kv: NULL argument");
  UA2_CHECK(kv->n_kv == kv->n_head && kv->head_size >= 16 && kv->head_size <= 256 && kv->head_size % 16 == 0,
            "ua2_qknorm_rope_kv: multi-head attention only (n_kv == n_head), head_size in 16..256");
  UA2_CHECK(kv->ring_pages == 0 || (kv->ring_pages & (kv->ring_pages - 1)) == 0, "ua2_qknorm_rope_kv: ring_pages=%d must be a power of two", kv->ring_pages);
  UA2_CHECK((q_norm_w == nullptr) == (k_norm_w == nullptr) && (!q_norm_w || (q_norm_b && k_norm_b)), "ua2_qknorm_rope_kv: norm weights come in pairs with biases");
  UA2_CHECK(rot_dim >= 0 && rot_dim <= kv->head_size && rot_dim % 2 == 0 && (rot_dim == 0 || (cos_t && sin_t)), "ua2_qknorm_rope_kv: bad rot_dim / tables");
  const size_t smem = (size_t)(2 * kv->head_size + 8) * sizeof(float);
  const dim3 grid((unsigned)R, kv->n_head), block(kv->head_size);
  if (dtype == UA2_BF16)
    hipLaunchKernelGGL((qknorm_rope_kv_kernel<UA2_BF16>), grid, block, smem, (hipStream_t)stream, qkv, row_pos, row_seq, q_norm_w, q_norm_b,
                       k_norm_w, k_norm_b, eps, cos_t, sin_t, rot_dim, q_out, *kv);
  else if (dtype == UA2_F32)
    hipLaunchKernelGGL((qknorm_rope_kv_kernel<UA2_F32>), grid, block, smem, (hipStream_t)stream, qkv, row_pos, row_seq, q_norm_w, q_norm_b,
                       k_norm_w, k_norm_b, eps, cos_t, sin_t, rot_dim, q_out, *kv);
  else {
    ua2_set_error("ua2_qknorm_rope_kv: bad dtype %d", dtype);
    return -1;
  }
  UA2_LAUNCH_CHECK();
  return 0;
}
