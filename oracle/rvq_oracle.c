/*
 * CPU ORACLE (test infrastructure, NOT product code): residual-VQ nearest-codeword search and lookup.
 *
 * Restates the algorithm of the reference's vendored RVQ
 *   tools/tokenizer/MimiCodec/model/quantization/core_vq.py
 *     EuclideanCodebook._quantize  :179-185   codes = argmin_c || x - e_c ||_2   (torch.cdist + argmin,
 *                                             first index wins ties, as torch.argmin)
 *     ResidualVectorQuantization.encode :365-376   residual -= e[codes]; next level
 *     ResidualVectorQuantization.decode :378-384   sum of the selected codewords, level by level
 * which is also the algorithm the live codec reaches through the un-vendored
 * vector_quantize_pytorch.ResidualVQ (ReasoningCodec_film/models/AudioDiffusion1D.py:388,529,535,544
 * quantise; :577-583 get_output_from_indices) — see SURVEY.md §8c: that package is absent, its parity
 * is "unpinned" and asserted by construction against this restatement.
 *
 * Arithmetic contract shared bit-for-bit with the HIP kernel (csrc/ua2_rvq.hip):
 *   d2(x, e) = sum_k (x_k - e_k)^2 accumulated in fp32 with one fused multiply-add per k, k ascending;
 *   argmin over c ascending with strict '<' (lowest index wins ties); residual and the quantised sum
 *   updated with plain fp32 subtract / add, level by level.
 * torch.cdist evaluates the same distance through a GEMM expansion, so indices can differ from the
 * reference only where two codewords are equidistant to within fp32 rounding; the golden test
 * (tests/test_oracle_rvq.py) measures that on vectors produced by the reference itself.
 */
#include <math.h>
#include <stdint.h>

/* x [N,D] fp32, emb [L,C,D] fp32 -> codes [N,L] int32, quantized [N,D] (sum of codewords, may be NULL),
 * margin [N,L] (second-best d2 minus best d2, may be NULL) */
void rvq_encode_oracle(const float* x, const float* emb, int64_t N, int L, int C, int D, int32_t* codes,
                       float* quantized, float* margin) {
  float res[1024];
  float q[1024];
  for (int64_t n = 0; n < N; ++n) {
    for (int k = 0; k < D; ++k) { res[k] = x[n * D + k]; q[k] = 0.0f; }
    for (int l = 0; l < L; ++l) {
      const float* cb = emb + (int64_t)l * C * D;
      float best = INFINITY, second = INFINITY;
      int bi = 0;
      for (int c = 0; c < C; ++c) {
        float acc = 0.0f;
        for (int k = 0; k < D; ++k) {
          const float d = res[k] - cb[(int64_t)c * D + k];
          acc = fmaf(d, d, acc);
        }
        if (acc < best) { second = best; best = acc; bi = c; }
        else if (acc < second) second = acc;
      }
      codes[n * L + l] = bi;
      if (margin) margin[n * L + l] = second - best;
      for (int k = 0; k < D; ++k) {
        const float e = cb[(int64_t)bi * D + k];
        res[k] = res[k] - e;
        q[k] = q[k] + e;
      }
    }
    if (quantized) for (int k = 0; k < D; ++k) quantized[n * D + k] = q[k];
  }
}

void rvq_decode_oracle(const int32_t* codes, const float* emb, int64_t N, int L, int C, int D, float* out) {
  for (int64_t n = 0; n < N; ++n)
    for (int k = 0; k < D; ++k) {
      float q = 0.0f;
      for (int l = 0; l < L; ++l) q = q + emb[((int64_t)l * C + codes[n * L + l]) * D + k];
      out[n * D + k] = q;
    }
}
