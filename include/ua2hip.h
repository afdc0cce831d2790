/*
 * ua2hip.h — C ABI of libua2hip.so: the MI355X (gfx950) kernels behind UniAudio 2.0's
 * audio-token generation hot path (BASELINE.json north_star).
 *
 * The reference (yangdongchao/UniAudio2) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md §8b): each entry point below replaces a *PyTorch call site* of the reference,
 * cited as file:line relative to the reference root.  The binding a maintainer of the
 * reference would add is a ctypes stub; uniaudio2_amd/_lib.py is that stub.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (void*), explicit sizes, a hipStream_t passed as void*.
 *   - every function returns 0 on success, negative on error; ua2_last_error() gives text.
 *   - the caller owns every buffer; the library owns only the handles it returns.
 *   - stream-ordered, no hidden synchronisation, no device allocation inside calls.
 *   - dtype = storage/compute type of weights, embedding tables and the KV cache:
 *       UA2_F32  exact-fp32 path (f32-input MFMA), for id-level parity with the fp32 reference
 *       UA2_BF16 bf16 operands, fp32 accumulate (MFMA 16x16x32 bf16), fp32 residual stream
 *     Activations between kernels are always fp32.
 */
#ifndef UA2HIP_H
#define UA2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UA2_VERSION 10

enum ua2_dtype { UA2_F32 = 0, UA2_BF16 = 1 };

/* A-operand producer fused into the GEMM (what the reference does right before the Linear). */
enum ua2_prologue {
  UA2_PRO_CAST = 0, /* x as is                                   (lit_model.py:511,595; model_new.py:617,631) */
  UA2_PRO_NORM = 1, /* RMSNorm(x)*w, fp32 math                    (lit_model.py:883-890 before :424 / :591)    */
  /* 2: reserved (round 1's merge of per-page attention partials; ua2_attn now writes the normalised row itself) */
  UA2_PRO_LOCAL_ATTN = 3, /* M == 1 only: the operand row IS the short-context attention of ua2_attn_local, computed
                       in the kernel (x = q [1, n_head*head_size], kv, row_pos, row_seq as for ua2_attn_local;
                       K == n_head*head_size).  Bit-identical to ua2_attn_local followed by UA2_PRO_CAST. */
  UA2_PRO_SCALED = 4  /* RMSNorm folded around the GEMM (UA2_BF16 only; the frame executor's form of lit_model.py:883-890
                       + :424 / :591):   y = rstd[m] * ( RNE_bf16(x (.) w) W^T ),   rstd = rsqrt(mean(x^2) + eps).
                       The operand RNE_bf16(x (.) w) is handed over by the PRODUCER of x (x_h: row-major bf16, launches of
                       one row tile; x_packed: fragment order, many-row launches) together with per-16-column sums of
                       squares of the fp32 x (x_ssq) — see y_norm_w below — so that no launch stands between the
                       producer and this GEMM, whatever the row count; the row scale is applied to the fp32 sums in
                       the epilogue.  Mathematically RMSNorm(x) W^T; numerically the bf16 rounding happens before
                       the row scale instead of after it (same relative error).  K % 32 == 0. */
};

/* What happens to the GEMM result (the ops the reference runs right after the Linear). */
enum ua2_epilogue {
  UA2_EPI_STORE = 0,    /* y = xW^T; optional per-tile (max,idx) partials for greedy sampling (model_new.py:146-187) */
  UA2_EPI_RESIDUAL = 1, /* y = resid + xW^T                          (lit_model.py:345,349)                    */
  UA2_EPI_SWIGLU = 2,   /* y = silu(xW1^T) * (xW2^T)                 (lit_model.py:592-594)                    */
  UA2_EPI_QKV_ROPE = 3, /* split q|k|v, RoPE on q,k (rope_mode), append k,v to the paged cache
                           (lit_model.py:431, 458-461, 778-807, 831-856; Moshi: transformer.py:386-396, rope.py:12-68) */
  UA2_EPI_GELU = 4      /* y = gelu(xW^T + b): exact erf form (Moshi FFN, transformer.py:559 F.gelu) or, with
                           act_kind = UA2_GELU_TANH, the tanh approximation (DiT FeedForward "gelu-approximate",
                           ReasoningCodec_film/models/model_config.json:4)                                        */
};

/* ua2_linear_args.act_kind: variant of the activation inside UA2_EPI_GELU / UA2_EPI_SWIGLU */
enum ua2_act_kind {
  UA2_ACT_DEFAULT = 0,          /* GELU: erf form; SWIGLU: silu(xW0^T) * (xW1^T)                                    */
  UA2_GELU_TANH = 1,            /* GELU: tanh approximation                                                         */
  UA2_GATE_SIGMOID_SECOND = 2   /* SWIGLU: (xW0^T + b0) * sigmoid(xW1^T + b1)  — the GLU of ReasoningCodec_film/
                                   modules/transformer.py:208-243 under power_normalized (proj output chunked x | gate) */
};

/* ua2_linear_args.sum_order */
enum ua2_sum_order { UA2_SUM_ORDER_INVARIANT = 0, UA2_SUM_ORDER_FREE = 1 };

/* flavours of UA2_PRO_NORM */
enum ua2_norm_kind {
  UA2_NORM_RMS_LIT = 0,   /* (x * rsqrt(mean(x^2) + eps)) * w           lit_model.py:883-890                     */
  UA2_NORM_RMS_MOSHI = 1, /* x * (alpha * rsqrt(eps + mean(x^2)))       llm_modules/transformer.py:34-46          */
  UA2_NORM_LAYERNORM = 2  /* (x - mean) * rsqrt(var + eps) * w + b      nn.LayerNorm (create_norm_fn :111-112)    */
};
/* RoPE flavours of UA2_EPI_QKV_ROPE */
enum ua2_rope_mode {
  UA2_ROPE_HALF_SPLIT = 0,  /* rotate-half (lit_model.py:795-806); weight packed with rope_head_size          */
  UA2_ROPE_INTERLEAVED = 1, /* adjacent pairs (2i, 2i+1) (rope.py:46-66); weight packed without permutation  */
  UA2_ROPE_NONE = 2         /* positional_embedding='none'                                                   */
};

const char* ua2_last_error(void);
int ua2_version(void);
/* sizeof() of the ABI structs as this library was compiled, for a binding to check its own layout against:
 * which = 0 ua2_kv_geom, 1 ua2_linear_args, 2 ua2_attn_args, 3 ua2_conv1d_args, 4 ua2_gpt_desc, 5 ua2_stage3_desc,
 * 6 ua2_convtc_args; else 0. */
size_t ua2_struct_size(int which);

/* Number of elements (of `dtype`) in the packed form of an [N,K] Linear weight. */
size_t ua2_packed_elems(int dtype, int64_t N, int64_t K);

/* Re-tile a Linear weight into MFMA B-fragment order so that every wave load is one
 * contiguous 1 KiB line burst: out[N/16][K/KC][64 lanes][16 B].  `src` is [N,K] row-major
 * (nn.Linear.weight, lit_model.py:356-360) or, with transposed=1, [K,N] (audio_head[i],
 * model_new.py:349,632).  src_dtype is the dtype of `src`; dst dtype is `dtype`
 * (fp32 -> bf16 rounds to nearest even, as torch .to(bfloat16)).
 * rope_head_size > 0 (fused qkv weights consumed by UA2_EPI_QKV_ROPE): the rows of every head are
 * additionally permuted so that output tile r of a head carries dims [8r,8r+8) and their half-split
 * rotation partners [hs/2+8r, hs/2+8r+8): RoPE then closes inside one 16-column tile. */
int ua2_pack_linear(const void* src, int src_dtype, int transposed, int64_t N, int64_t K,
                    void* out, int dtype, int rope_head_size, void* stream);

/* Geometry of the paged KV cache shared by the QKV epilogue and the attention kernel.
 * Pool layout (per layer): [n_pages][n_kv][UA2_PAGE][head_size] of `dtype`, K and V separate.
 * page_table[seq*max_pages + p] = page id holding positions [p*UA2_PAGE, (p+1)*UA2_PAGE). */
#define UA2_PAGE 64

typedef struct ua2_kv_geom {
  void* k_pool;
  void* v_pool;
  const int32_t* page_table;
  int32_t max_pages; /* per sequence */
  int32_t n_kv;      /* n_query_groups */
  int32_t n_head;
  int32_t head_size;
  int32_t ring_pages; /* 0: linear cache (page of position p = p / UA2_PAGE).  > 0: ring cache — position p lives in table column
                         (p / UA2_PAGE) % ring_pages (ring_pages a power of two), so a streaming session of any length re-uses ring_pages pages per sequence
                         (RingKVCache, llm_modules/transformer.py:211-278).  The caller guarantees ring_pages * UA2_PAGE exceeds
                         the attention window plus the positions written per launch; ua2_attn needs window > 0 with it. */
} ua2_kv_geom;

typedef struct ua2_linear_args {
  int32_t dtype, prologue, epilogue;
  int32_t M, N, K;        /* rows, out features, in features */
  const float* x;         /* CAST/NORM: [M, ldx] fp32 */
  int32_t ldx;
  const float* norm_w;    /* NORM: [K] fp32 */
  float eps;
  const void* w0;         /* packed weight */
  const void* w1;         /* SWIGLU: packed fc_2 */
  float* y;               /* STORE/RESIDUAL/SWIGLU: [M, ldy] fp32 (STORE: may be NULL if only partials wanted) */
  int32_t ldy;
  const float* resid;     /* RESIDUAL: [M, ldr], may alias y */
  int32_t ldr;
  float* part_max;        /* STORE, optional: [M, ceil(N/16)] */
  int32_t* part_idx;
  const int32_t* forbid;  /* optional [M]: columns < forbid[m] are excluded from the partial arg-max */
  const int32_t* row_pos; /* QKV_ROPE, LOCAL_ATTN: [M] absolute position of each row */
  const int32_t* row_seq; /* QKV_ROPE: [M] sequence (page-table row) of each row */
  const float* rope_cos;  /* QKV_ROPE: [max_pos, head_size/2] */
  const float* rope_sin;
  float* q_out;           /* QKV_ROPE: [M, n_head*head_size] fp32, rotated */
  ua2_kv_geom kv;         /* QKV_ROPE, LOCAL_ATTN */
  const float* norm_b;    /* NORM, LAYERNORM only: [K] bias */
  int32_t norm_kind;      /* enum ua2_norm_kind */
  const float* out_scale; /* RESIDUAL, optional [N]: y = resid + out_scale[n] * xW^T  (LayerScale, transformer.py:97) */
  int32_t rope_mode;      /* enum ua2_rope_mode */
  void* workspace;        /* optional device scratch of >= ua2_linear_workspace_bytes(dtype, M, K): enables the
                             large-M kernel (normalised operand rows packed once, 128-row tiles).  Results are
                             bit-identical with or without it; without it M > 16 streams the weights once per
                             16-row tile. */
  size_t workspace_bytes;
  /* Producer / consumer hand-over of the many-row operand, skipping the consumer's prep launch:
     y_packed (UA2_EPI_SWIGLU, UA2_EPI_GELU; N % chunk == 0): the result, rounded to `dtype`, also (or, with y == NULL, only) in the
       packed operand layout [ceil(M/16)][N/KC][64 lanes][16 B] of a following K = N launch;
     x_packed (UA2_PRO_CAST, M spanning more than one row tile): the operand already in that layout (K = this launch's K).
     Same bits as the unpacked route: the packed value is the same RNE cast the prep launch applies. */
  void* y_packed;
  const void* x_packed;
  const float* bias;      /* optional [N]: nn.Linear bias added to xW^T before the epilogue's activation / residual /
                             split (NULL for every Linear of the LM).  With UA2_EPI_QKV_ROPE only for weights packed
                             WITHOUT rope_head_size (rope_mode INTERLEAVED / NONE), where column n is source row n */
  const float* bias1;     /* SWIGLU: bias of w1 */
  int32_t act_kind;       /* enum ua2_act_kind */
  /* Scaled-norm hand-over (UA2_BF16).  Producer side — UA2_EPI_RESIDUAL / UA2_EPI_STORE with y_norm_w != NULL: besides
     y the launch writes, for the UA2_PRO_SCALED consumer that follows, RNE_bf16(y[m][n] * y_norm_w[n]) into y_h (row-major
     [M, ldh] bf16) and / or y_packed (fragment order, N % 32 == 0), and y_ssq[m][n / 16] = the sum of y[m][n]^2 over the
     16-column tile, added in a fixed butterfly order (xor 1, 2, 4, 8; the first level fused: an even column c contributes
     fma(y_c, y_c, RN(y_{c+1}^2))) — the same tree in every kernel, so a row's statistic does not depend on the row count.  N % 32 == 0 (the launcher enforces it: whole bf16 MFMA chunks of the consumer).
     Consumer side — UA2_PRO_SCALED: x_h (M <= the decode kernel's row tile) or x_packed, and x_ssq [M, K / 16]; eps as
     for UA2_PRO_NORM.  rstd[m] = rsqrt(sum_j x_ssq[m][j] / K + eps), j summed as 16 interleaved chains + butterfly. */
  const float* y_norm_w;
  void* y_h;
  int32_t ldh;            /* row stride (elements) of y_h / x_h */
  float* y_ssq;
  const void* x_h;
  const float* x_ssq;
  /* [v7] Optional scratch for a K split of UA2_EPI_RESIDUAL launches that take the tiled many-row kernel with a long K and a
     grid too small for the device (the codec DiT's FF2: 1000 x 1536, K = 6144 on 192 workgroups).  When given (and N % 64 == 0,
     no hand-over outputs, K >= 4096), the launcher may cut K into S <= 4 slabs — S a function of (M, N, K) AND of how many slabs
     split_ws_bytes holds: S = min(4, ceil(768 / workgroups(M, N)), split_ws_bytes / (M N 4)); a shorter last window of the DiT or a
     smaller scratch therefore means another S, i.e. another (equally valid) summation order — whose
     workgroups run side by side and write fp32 partial sums [S][M][N] here; a second launch forms
     y = resid + out_scale (.) ((((s0 + s1) + s2) + s3) + bias) in that fixed order.  Deterministic, but the last bits differ
     from the unsplit launch (one chain per slab instead of the decode kernel's ranges): callers that rely on the row-count
     invariance of ua2_linear (every launch of the LM) leave it NULL.  Needs S * M * N * 4 bytes (4 * M * N * 4 always suffices). */
  float* split_ws;
  size_t split_ws_bytes;
  /* [v8] Summation-order contract of the K sum (enum ua2_sum_order).  UA2_SUM_ORDER_INVARIANT (0, the default): a row's bits are
     those of the decode kernel whatever the row count or the kernel (`waves` partial chains over fixed K ranges, added in wave
     order) — every launch of the LM's fp32 plan and of its bf16 plan below 2048 rows.  UA2_SUM_ORDER_FREE: the order is the
     launcher's choice (one chain over K per slab on the 256-row-tile kernel of csrc/ua2_gemm2.hip; S slabs with split_ws) —
     deterministic for a given (M, N, K, scratch), fp32 rounding noise (~1e-6 relative) against the invariant form; for callers
     outside the LM's row-invariance contract (codec DiT, AudioThinking, Mimi) and, as a plan option, LM launches of >= 2048
     rows.  bf16 only; launches outside the fast kernel's forms silently take the invariant kernels. */
  int32_t sum_order;
  /* [v9] LayerNorm hand-over (UA2_SUM_ORDER_FREE, UA2_EPI_RESIDUAL, UA2_BF16; the codec DiT: attention.py:311-319 / :388-390 behind
     :345-349 / :401-405).  With y_ln_w != NULL the launch also writes, for the K = N GEMM that follows,
         y_packed = RNE_bf16( (y[m] - mean_m) * rstd_m * y_ln_w + y_ln_b ),   mean / rstd over the N columns of y[m] (eps = y_ln_eps),
     in fragment order — F.layer_norm(y) * w + b, the operand its consumer's UA2_PRO_NORM prep launch would build (the consumer then
     takes it as x_packed with UA2_PRO_CAST).  Where the launch runs as K slabs the combine forms it in the same pass (one launch
     instead of combine + prep); otherwise a row pass follows the GEMM.  Only the order-free kernel implements it (N % 4 == 0,
     N <= 2048): ua2_linear_order_free_accepts() tells a caller beforehand whether a launch will be taken; a launch that carries
     y_ln_w and is not is an error.  The statistics are summed in this kernel's own order (fp32): same value as the prep launch's to
     rounding, not to the bit — the order-free contract. */
  const float* y_ln_w;
  const float* y_ln_b;
  float y_ln_eps;
  /* [v10] Optional scratch for the RANGE split of row-invariant launches (UA2_SUM_ORDER_INVARIANT, UA2_BF16, UA2_EPI_RESIDUAL, K = 8192,
     33-64 rows: the batched-decode down-projections, lit_model.py:591-595).  The decode kernel adds `waves` partial chains over fixed K
     ranges in range order; with range_ws a launch may run each range on its own workgroups (the range's operand staged once per
     workgroup, wide column groups) and add the partials [waves][ceil(M/16)][ceil(N/16)][256] in a second launch IN THAT ORDER from zero —
     the same sums, the same bits as without it (tests/test_gpu_invariance.py).  Needs waves * ceil(M/16)*16 * ceil(N/16)*16 * 4 bytes
     (16 ranges: 12.6 MB at 64 x 3072), 16-byte aligned; NULL = never.  Measured slower than the one-launch form in the LM's frame
     (profiles/r6_range_split.txt: 15.3 + 5.9 us against 20.4), so the frame executor leaves it NULL unless UA2_RANGE_SPLIT=1. */
  float* range_ws;
  size_t range_ws_bytes;
} ua2_linear_args;

int ua2_linear(const ua2_linear_args* a, void* stream);
/* [v9] 1 when a launch with these arguments (sum_order = UA2_SUM_ORDER_FREE) will run on the order-free kernel (csrc/ua2_gemm2.hip) as it
 * stands — shape, alignment, the tile-count rule, scratch for K slabs; 0 when ua2_linear would take the row-invariant kernels instead.
 * Nothing is launched. */
int ua2_linear_order_free_accepts(const ua2_linear_args* a);
/* Bytes of ua2_linear_args.workspace that a launch with these M, K needs (the operand rows in MFMA
 * fragment order, rows padded to 16, K padded to the chunk size). */
size_t ua2_linear_workspace_bytes(int dtype, int64_t M, int64_t K);
/* Test hook, returns the previous setting.  0 (default): the launcher picks — row-tiled decode-regime
 * kernel for one row tile, large-M kernels when a workspace is supplied and M spans more than one.
 * 1: removed (round 1's general-M kernel), treated as 0.
 * 2: never the large-M kernels.  3: a large-M kernel whenever a workspace is supplied, even for M = 1
 * (4: always its skinny form, 5: always its 128-row tiled form).
 * Modes 0, 2, 3, 4 and 5 produce bit-identical results (tests/test_gpu_invariance.py). */
int ua2_debug_force_general_linear(int on);

/* Test hooks (ABI v9).  ua2_debug_kernel_launches: how many launches of a kernel family this process has issued so far —
 * "gemm2" (ua2_gemm2.hip, the order-free many-row GEMM), "gemm" (ua2_gemm.hip's tiled kernel), "skinny2", "gemv", "rsplit" ([v10] the range split
 * of ua2_skinny.hip: main + combine count once); -1 for an unknown
 * name.  A test that claims "the order-free kernel ran" reads the counter on both sides of the call instead of trusting the
 * launcher's rules.  ua2_debug_refresh_env: the launchers read their UA2_* tuning / A-B environment variables ONCE (they used to
 * call getenv on every launch); a process that changes one of them afterwards (the tests do) calls this to have them read again. */
int64_t ua2_debug_kernel_launches(const char* family);
void ua2_debug_refresh_env(void);

/* Measurement helper (bench.py roofline leg): launches args[0..n) back to back `iters` times on
 * `stream`, bracketed by hipEvents recorded on that same stream, waits for the stop event and
 * returns the elapsed milliseconds in *ms_out.  No other work is enqueued in between. */
int ua2_linear_chain_timed(const ua2_linear_args* args, int32_t n, int32_t iters, void* stream, float* ms_out);

/* Decode/prefill attention over the paged cache: one query row per (row, head); one workgroup per (row, kv-head)
 * walks positions 0..row_pos (waves own contiguous ranges, online softmax, fixed-order merge in LDS) and writes the
 * normalised output.  Replaces repeat_interleave + masked SDPA (lit_model.py:478-481, 529-531): GQA without
 * materialising K/V per query head, causal mask = "positions <= row_pos". */
typedef struct ua2_attn_args {
  int32_t dtype;
  int32_t R;              /* query rows */
  const float* q;         /* [R, n_head*head_size] fp32 */
  const int32_t* row_pos; /* [R] */
  const int32_t* row_seq; /* [R] page-table row of each query row; NULL: row r is sequence r */
  ua2_kv_geom kv;
  float* y;               /* [R, n_head*head_size] fp32 normalised output (may be NULL when y_packed is given) */
  int32_t window;         /* > 0 = attend only to the last `window` positions (Moshi `context`,
                             transformer.py:405-406: delta < context); 0 = all positions <= row_pos */
  void* y_packed;         /* optional: the output rounded to `dtype` in the packed operand layout of the
                             O-projection (ua2_linear_args.x_packed), K = n_head*head_size */
  /* Grouped form (prefill of prompts, dense encoders, the DiT): when group_rows != NULL and dtype == UA2_BF16 the rows
     are processed by an MFMA flash kernel with LDS-staged K/V pages instead of row by row.  The host lists, per group,
     up to group_q_tiles * 16 query rows OF ONE SEQUENCE (row indices into q / row_pos, -1 = padding); rows outside every
     group are not computed.  Same arithmetic contract (bf16 K/V, fp32-grade q and softmax), a row's result independent
     of how rows are grouped. */
  const int32_t* group_rows;   /* [n_groups, group_q_tiles * 16] device */
  const int32_t* group_seq;    /* [n_groups] page-table row of the group's sequence */
  const int32_t* group_nkeys;  /* [n_groups] 1 + the largest row_pos in the group (keys visited: 0 .. nkeys-1) */
  int32_t n_groups;
  int32_t group_q_tiles;       /* 16-row tiles per group: 2 with grouped-query heads (n_head > n_kv), 4 with n_head == n_kv */
  int32_t flags;               /* [v8] UA2_ATTN_BF16_QP: the grouped form may round q and the softmax weights to bf16 once (no hi / lo
                                  split) — what torch SDPA under bf16 autocast computes (the codec's DiT, reason_tokenizer.py:265); half the
                                  matrix work.  Served at head size 64, 8 query tiles; ignored (the fp32-grade form runs) elsewhere. */
} ua2_attn_args;
#define UA2_ATTN_BF16_QP 1

int ua2_attn(const ua2_attn_args* a, void* stream);
/* Short-context form for the local (depth) decoder (model_new.py:629-641): every row attends to positions
 * 0..row_pos[r], row_pos[r] < 8, all in the first cache page of its sequence.  Uses q, row_pos, row_seq, kv, y
 * of ua2_attn_args ([R, n_head*head_size] fp32 out); exact-operation softmax in position order. */
int ua2_attn_local(const ua2_attn_args* a, void* stream);

/* Producer half of the scaled-norm hand-over (see ua2_linear_args.y_norm_w) for the row-wise kernels that feed a GPT's first
 * layer: besides its fp32 output `o` [M, C] the kernel writes RNE_bf16(o * norm_w) into h (row-major, stride ldh) and / or
 * packed (fragment order), and ssq [M, C / 16] (the same butterfly tree as the linear epilogues).  UA2_BF16; C % 32 == 0. */
typedef struct ua2_handover {
  const float* norm_w;   /* [C] weight of the RMSNorm the consumer folds */
  void* h;               /* [M, ldh] bf16, or NULL */
  int32_t ldh;
  void* packed;          /* [ceil(M/16)][C/32][64 lanes][16 B], or NULL */
  float* ssq;            /* [M, C / 16] */
} ua2_handover;

/* Frame embedding (model_new.py:594-600, 604, 665-673): for each row,
 * audio_sum = sum_i mask[i] * audio_emb[tok[i] + i*V_a]  (i = 0..n_cb-1, in order), text = wte[tok[n_cb]].
 * ho (optional): hand-over of audio_sum to the understanding expert's first layer. */
int ua2_embed_frame(int dtype, int32_t M, int32_t C, int32_t n_cb, int32_t va,
                    const int32_t* tokens /* [M, n_cb+1] */, const uint8_t* mask /* [M, n_cb+1] */,
                    const void* audio_emb, const void* wte, float* audio_sum /* [M,C] */, float* text /* [M,C] */,
                    const ua2_handover* ho, void* stream);

/* Final RMSNorm of a GPT + the step-mask blends (lit_model.py:164; model_new.py:607,610,613):
 *   n = RMSNorm(x)*w ; out1 = n*fa + other*fb ; out2 = n (optional)
 * fa = mask[m, col_a] if col_a >= 0 else 1 ; fb = mask[m, col_b] if other else 0. */
int ua2_rmsnorm_blend(int32_t M, int32_t C, const float* x, const float* w, float eps,
                      const float* other, const uint8_t* mask, int32_t mask_ld, int32_t col_a, int32_t col_b,
                      float* out1, float* out2, const ua2_handover* ho /* optional: hand-over of out1 */, void* stream);

/* Greedy sampling tail (model_new.py:146-187 with topk=1; lowest index wins ties) fused with
 * the next-step embedding gather (model_new.py:640,662-663):
 *   tok = argmax over the per-tile partials; out_tokens[m*out_ld + out_col] = tok;
 *   if emb: next_h[m,:] = emb[tok + emb_row_offset, :]  */
int ua2_argmax_embed(int dtype, int32_t M, int32_t n_part, const float* part_max, const int32_t* part_idx,
                     int32_t* out_tokens, int32_t out_ld, int32_t out_col,
                     const void* emb, int32_t emb_row_offset, int32_t C, float* next_h, void* stream);

/* Classifier-free guidance (model_new.py:618-622, 634-637) over `pairs` (conditional, unconditional) pairs of logit rows
 * (rows 2p, 2p + 1 of [2 pairs, ld] fp32; the reference has one pair, batched generation has one per utterance):
 * guided = l1 + (l0 - l1) * scale, written to BOTH rows of the pair, and the per-16-column arg-max partials of both rows
 * ([2 pairs, ceil(V/16)], same format and tie rule as UA2_EPI_STORE; columns < forbid[2p] excluded) rebuilt from it, so
 * ua2_argmax_embed / ua2_sample_topk run unchanged and both rows continue from the same token. */
int ua2_cfg_mix(float* logits, int32_t ld, int32_t V, float scale, const int32_t* forbid, float* part_max,
                int32_t* part_idx, int32_t pairs, void* stream);

/* Top-k sampling tail (model_new.py:146-187 with topk > 1: temperature, forbid_prefix, keep logits >= the
 * k-th largest, exponential-race multinomial draw) + next-step embedding gather.  Philox4x32-10 keyed by
 * (seed + device seed word, draw index, row, stream_id) where `counter` is a DEVICE int32[3]: [0] = draw index,
 * [1], [2] = low / high half of a 64-bit word added to `seed` (zeros for a purely by-value seed; the frame
 * executor keeps its seed there so that one captured graph serves every seed).  Reproducible under graph replay;
 * it does not reproduce torch's generator stream (parity with the reference is distributional).
 * row_key_shift (0 or 1): the row part of the key is m >> row_key_shift — 1 makes the two rows of a guidance pair draw the
 * same numbers (they hold the same guided logits: both continue from one sample, model_new.py:622). */
int ua2_sample_topk(int dtype, int32_t M, const float* logits, int32_t ld, int32_t V, int32_t topk, float temperature,
                    const int32_t* forbid, uint64_t seed, const int32_t* counter, int32_t stream_id,
                    int32_t* out_tokens, int32_t out_ld, int32_t out_col, const void* emb, int32_t emb_row_offset,
                    int32_t C, float* next_h, int32_t row_key_shift, void* stream);

/* ---- codec: residual vector quantisation ------------------------------------------------ */

/* Nearest-codeword search, level by level on the residual (core_vq.py:179-185, 365-376; the live codec's
 * ResidualVQ calls AudioDiffusion1D.py:388,529,535,544).  x [N,D] fp32 in codebook space (after any
 * project_in), emb [L,C,D] fp32, embT [L,D,C] = the same codebooks k-major (coalesced scan).
 * codes [N,L] int32; quantized [N,D] = sum of the chosen codewords (may be NULL).
 * d2 = sum_k fma(x_k-e_k, x_k-e_k, .), k ascending; lowest index wins ties (= oracle/rvq_oracle.c bit for bit).
 * workspace (optional, >= ua2_rvq_workspace_bytes(N, L) bytes of device scratch): lets a launch with few vectors (one
 * clip) split each level's codebook over several workgroups per group of 8 vectors (candidates merged by 64-bit atomic
 * min, lowest index on ties); same codes and sums with or without it.  The workgroups of a split launch wait for each
 * other with BOUNDED spins; if one expires (a peer held back by another stream / process) the launch raises a device
 * flag and a second, self-contained launch behind it on the same stream — a no-op when the flag is clean — recomputes
 * every vector: rc 0 always means oracle-exact codes.  ua2_rvq_fallbacks counts how often that happened (health counter;
 * synchronous copy from the device).  The second launch is enqueued on EVERY split encode (its workgroups read the flag and
 * return when it is clean: ~2 us of launch + boundary per encode against the 51 us search — measured, profiles/r3_rvq.txt — the
 * price of never needing a host round trip to learn whether the spin expired).  Env UA2_RVQ_SPIN_LIMIT overrides the spin bound (tests force the path with 0). */
int ua2_rvq_encode(const float* x, const float* emb, const float* embT, int64_t N, int32_t L, int32_t C, int32_t D,
                   int32_t* codes, float* quantized, void* workspace, size_t workspace_bytes, void* stream);
size_t ua2_rvq_workspace_bytes(int64_t N, int32_t L);
int ua2_rvq_fallbacks(uint32_t* out);
/* Lookup + sum over levels (core_vq.py:378-384; AudioDiffusion1D.py:577-583 get_output_from_indices). */
int ua2_rvq_decode(const int32_t* codes, const float* emb, int64_t N, int32_t L, int32_t C, int32_t D, float* out,
                   void* stream);

/* ---- codec: 1-D convolutions -------------------------------------------------------------- */

enum ua2_act { UA2_ACT_NONE = 0, UA2_ACT_PRELU = 1, UA2_ACT_ELU = 2, UA2_ACT_TANH = 3, UA2_ACT_ROUND9 = 4 };

/* out[b][co][t] = post( bias[co] + sum_{ci,j} W[co][ci][j] * pre(x[b][ci][(t*stride + j*dilation - pad_left)/in_repeat]) )
 *                 (+ residual[b][co][t]),   zero outside [0, Tin*in_repeat).
 * Covers scalar24k.py Conv1d :36-74 (causal: pad_left = d(k-1); else symmetric) and conv.py StreamingConv1d
 * :232-254 (pad_left = k_eff - stride, right pad implied by Tout); PReLU/ELU/tanh/round9 fused before or
 * after; in_repeat = repeat-upsampling (scalar24k.py:136-138).
 * Transposed conv (scalar24k.py:76-112, conv.py:306-329): out_phases = stride P, `w` holds the P phase
 * filters stacked (rows phase*Cout + co, taps of a phase in descending order: ua2 host helper
 * pack_convtr_weight), K = taps per phase, pad_left = K - 1, stride = dilation = 1; result written to
 * y[co][t*P + phase - out_trim_left], Tout = final (trimmed) length.
 * w: ua2_pack_linear(fp32) of the [rows, Cin_pad*K] matrix, Cin padded with zeros to a multiple of 16.
 * precision = 1 ("bf16 x 3"): every fp32 operand is split into two bf16 halves and the product taken as
 * Wh Xh + Wh Xl + Wl Xh on the bf16 MFMA with fp32 accumulation (~2^-16 relative per product instead of exact; 5x the
 * matrix rate).  w / w_lo then hold the hi / lo halves, each ua2_pack_linear(bf16) of the [rows, G*K*32] matrix whose
 * reduction index is (channel group of 32, tap, channel in group), Cin zero-padded to a multiple of 32 (host helper
 * ops.pack_conv_weight_x3). */
typedef struct ua2_conv1d_args {
  int32_t B, Cin, Cout, Tin, Tout;
  int32_t K, stride, dilation, pad_left;
  int32_t in_repeat, out_phases, out_trim_left;
  int32_t pre_act, post_act;       /* enum ua2_act */
  const float* x;                  /* [B, Cin, Tin] fp32 */
  const void* w;                   /* packed fp32 */
  const float* bias;               /* [Cout] or NULL */
  const float* pre_alpha;          /* PReLU slope for pre_act (1 value) or NULL */
  const float* post_alpha;         /* PReLU slope(s) for post_act */
  int32_t post_alpha_n;            /* 1 or Cout */
  const float* residual;           /* [B, Cout, Tout] or NULL, added after post_act (scalar24k.py:151) */
  float* y;                        /* [B, Cout, Tout] */
  const void* w_lo;                /* precision 1: packed bf16 low halves of the filter */
  int32_t precision;               /* 0 = exact fp32 (f32-input MFMA), 1 = bf16 x 3 */
  /* precision 1, optional: fused residual unit (scalar24k.py:143-151) — y = residual + PReLU(alpha2, W2 h + bias2) with
     h = post_act(conv(x) + bias) kept on chip; W2 a 1 x 1 conv [Cout, Cout] packed like w / w_lo (pack_conv_weight_x3 of
     [Cout, Cout, 1]).  Needs Cin == Cout in {32, 64, 128}, stride 1, Tin == Tout, residual. */
  const void* w2;
  const void* w2_lo;
  const float* bias2;              /* [Cout] or NULL */
  const float* alpha2;             /* PReLU slope (1 value) of the second activation, NULL = 0 (ReLU) */
} ua2_conv1d_args;

int ua2_conv1d(const ua2_conv1d_args* a, void* stream);
/* ---- codec decode side: convolutions on "time-major split planes" (round 4) --------------------------------------
 * Between the layers of the waveform DECODER (scalar24k.py:403-407: ScalarModel.decode = Conv1d -> N x ResDecoderBlock
 * [ConvTranspose1d up-sampler -> 5 dilated ResidualUnits, :143-151] -> PostProcessor -> Conv1d) activations travel as two
 * bf16 planes hi = RNE(x), lo = RNE(x - hi) (16 significant bits, the same 4 bytes per element as fp32), each laid out
 * [B][T][C] — time-major, channels contiguous — instead of fp32 [B][C][T].  The layout is what the bf16 x 3 matrix-pipe
 * form consumes (an MFMA B fragment = 8 consecutive channels of one time step), so a layer's input window goes from L2
 * to LDS by LDS-DMA (global_load_lds, no register staging, no conversion, no transposing LDS writes) and the epilogue of
 * every layer emits the next layer's operand.  C must be a multiple of 32 (channel groups of the MFMA K dimension).
 * Arithmetic per output: the ua2_conv1d precision-1 sum (chunks = (channel group, tap) ascending; per chunk Wl*Xh, Wh*Xl,
 * Wh*Xh), bias added once, PReLU with single roundings, residual x = hi + lo (exact in fp32) added last, then the hi / lo
 * split.  Three kernels with identical bits (`variant`): 1 = plain (any K <= 32, reference of the bit-identity test),
 * 2 = software-pipelined LDS-DMA form (K in {1, 2, 7}), 3 = big-tile form (fused residual units of 32 / 64 channels: one 512- / 1024-step
 * tile per 8-wave workgroup), 0 = automatic (big-tile for those units, pipelined otherwise, plain as the fallback). */
typedef struct ua2_convtc_args {
  int32_t B, Cin, Cout, Tin, Tout;
  int32_t K, dilation, pad_left;   /* stride is 1 on the decode side */
  int32_t in_repeat;               /* input row of window position p is p / in_repeat (repeat-upsampling, scalar24k.py:136-140) */
  int32_t out_phases, out_trim_left; /* transposed conv as out_phases phase filters: packed row n = phase * Cout + co lands at
                                      y[t * out_phases + phase - out_trim_left][co] (see ua2_conv1d) */
  int32_t post_act;                /* UA2_ACT_NONE or UA2_ACT_PRELU */
  int32_t variant;                 /* 0 auto (fused 32- / 64-channel residual units: the big-tile kernel; everything else the pipelined one,
                                      the plain one for shapes outside both), 1 plain, 2 pipelined, 3 big-tile (error if the shape is outside
                                      the requested kernel's instantiations).  All three give identical bits. */
  const uint16_t* x_hi;            /* [B, Tin, Cin] bf16 */
  const uint16_t* x_lo;
  const void* w;                   /* ops.pack_conv_weight_x3 of the [rows, Cin, K] filter: hi ... */
  const void* w_lo;                /* ... and lo halves */
  const float* bias;               /* [Cout] or NULL */
  const float* post_alpha;         /* PReLU slope(s) */
  int32_t post_alpha_n;            /* 1 or Cout */
  /* fused residual unit (scalar24k.py:143-151): y = x + PReLU(alpha2, W2 h + bias2), h = post_act(conv(x) + bias) kept on
     chip; needs Cin == Cout in {32, 64, 128}, out_phases == 1, in_repeat == 1, Tin == Tout.  The residual is the input. */
  const void* w2;
  const void* w2_lo;
  const float* bias2;
  const float* alpha2;             /* 1 value; NULL = slope 0 */
  /* separate residual (the un-fused second conv of a wide residual unit): added after the activation */
  const uint16_t* res_hi;          /* [B, Tout, Cout] bf16 or NULL */
  const uint16_t* res_lo;
  /* output: planes, or fp32 [B, Cout, Tout] (the waveform: last layer) — exactly one of (y_hi & y_lo) / y_f32 */
  uint16_t* y_hi;
  uint16_t* y_lo;
  float* y_f32;
} ua2_convtc_args;

int ua2_conv1d_tc(const ua2_convtc_args* a, void* stream);
/* fp32 [B, C, T] -> planes [B, T, C] (hi, lo) and back (x = hi + lo, exact).  C % 2 == 0. */
int ua2_tc_pack(const float* x, uint16_t* hi, uint16_t* lo, int32_t B, int32_t C, int32_t T, void* stream);
int ua2_tc_unpack(const uint16_t* hi, const uint16_t* lo, float* y, int32_t B, int32_t C, int32_t T, void* stream);

/* torch.nn.AvgPool1d(kernel_size=k) over the last axis of [rows, Tin] (scalar24k.py:118). */
int ua2_avgpool1d(const float* x, float* y, int64_t rows, int32_t Tin, int32_t k, void* stream);

/* Depthwise (groups == channels) 1-D convolution / transposed convolution, exact fp32, taps in ascending order:
 *   conv:   y[b,c,t] = bias[c] + sum_j w[c,j] * x[b,c, t*stride + j*dilation - pad_left]      (zero outside)
 *   convtr: y[b,c,t] = bias[c] + sum_{ti*stride + j == t + pad_left} w[c,j] * x[b,c,ti]        (pad_left = left trim)
 * Replaces the channel-wise resamplers of tools/tokenizer/MimiCodec/model/modules/resample.py:13-119
 * (nn.Conv1d / nn.ConvTranspose1d with groups = dimension). x [B,C,Tin], w [C,K], y [B,C,Tout]. */
int ua2_dwconv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t C, int32_t Tin,
                 int32_t Tout, int32_t K, int32_t stride, int32_t dilation, int32_t pad_left, int32_t transposed,
                 void* stream);

/* ---- codec: glue of the neural stages (flow-matching DiT, AudioThinking encoder) ------------------------------- */

enum ua2_ew_act { UA2_EW_IDENTITY = 0, UA2_EW_SILU = 1, UA2_EW_SIGMOID = 2, UA2_EW_TANH = 3 };

/* out[i] = alpha * a[i % na] * (b ? b[i % nb] : 1) + (c ? c[i % nc] : 0) + beta, i < n (modulo = broadcast of a shorter
 * operand over rows).  The broadcast multiply-adds of the DiT and the Euler solver: adaLN modulation vectors
 * (models/attention.py:308-311), gated residuals (:345-349, 401-405), ProjectLayer's k^-0.5 (transformer_1d_flow.py:31),
 * guidance and the Euler step (AudioDiffusion1D.py:104,116-123).  out may alias a, b or c. */
int ua2_ew_fma(float* out, int64_t n, const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc,
               float alpha, float beta, void* stream);
/* out[i] = act(x[i]) (SiLU of adaLN-single, transformer_1d_flow.py:113; nn.SiLU of TimestepEmbedding). */
int ua2_ew_act(float* out, const float* x, int64_t n, int32_t act, void* stream);
/* out[r, :] = in[idx[r], :], zeros where idx[r] < 0; C % 4 == 0.  Nearest-neighbour interpolation along time
 * (AudioDiffusion1D.py:450,512,590) and the cls-token interleave / extraction (:458-486) with host-built index lists. */
int ua2_gather_rows(float* out, const float* in, const int32_t* idx, int64_t R, int32_t C, void* stream);
/* time_film (AudioDiffusion1D.py:428-438): params [R, 2C] = (delta_gamma | beta); out = (1 + gamma_scale * tanh(dg)) * x
 * + beta; rows of a batch element with batch_mask[b] != 0 (the reference's `torch.rand(B,1,1) < 0.2` draw, passed in so
 * that runs are reproducible) pass x through unchanged.  R = B * rows_per_batch. */
int ua2_time_film(float* out, const float* params, const float* x, const uint8_t* batch_mask, int64_t R,
                  int32_t rows_per_batch, int32_t C, float gamma_scale, void* stream);
/* F.layer_norm over the last axis of [R, C]; w, b optional (NULL = elementwise_affine=False, transformer_1d_flow.py:252). */
int ua2_layernorm_rows(float* out, const float* x, const float* w, const float* b, int64_t R, int32_t C, float eps, void* stream);
/* q/k LayerNorm over the head dim + partial rotary embedding + K/V append to the paged cache for the x-transformers
 * style attention of the AudioThinking encoder (modules/transformer.py:447-485, 146-170).  qkv [R, 3*n_head*hs] =
 * (q | k | v); q_out [R, n_head*hs] fp32; cos_t / sin_t [max_pos, rot_dim/2]; norm weights NULL = no q/k norm;
 * rot_dim 0 = no rotary.  Multi-head only (kv->n_kv == kv->n_head). */
int ua2_qknorm_rope_kv(int dtype, const float* qkv, int64_t R, const int32_t* row_pos, const int32_t* row_seq,
                       const float* q_norm_w, const float* q_norm_b, const float* k_norm_w, const float* k_norm_b, float eps,
                       const float* cos_t, const float* sin_t, int32_t rot_dim, float* q_out, const ua2_kv_geom* kv, void* stream);

/* ---- whole-frame executor -------------------------------------------------------------- */

typedef struct ua2_gpt_desc {
  int32_t n_layer, n_embd, n_head, n_kv, head_size, inter;
  float eps;
  /* per layer arrays of device pointers (host arrays of length n_layer) */
  const void* const* qkv;    /* packed [ (n_head+2*n_kv)*hs, n_embd ] */
  const void* const* proj;   /* packed [ n_embd, n_head*hs ] */
  const void* const* fc1;    /* packed [ inter, n_embd ] */
  const void* const* fc2;
  const void* const* mlp_proj; /* packed [ n_embd, inter ] */
  const float* const* norm1;
  const float* const* norm2;
  const float* ln_f;
  const float* rope_cos;     /* [max_pos, hs/2] */
  const float* rope_sin;
  void* const* k_pool;       /* per layer */
  void* const* v_pool;
  const int32_t* page_table;
  int32_t max_pages;
} ua2_gpt_desc;

typedef struct ua2_stage3_desc {
  int32_t dtype;
  int32_t n_cb;         /* audio_num_codebooks (8) */
  int32_t va;           /* audio_semantic_vocab_size + audio_reason_vocab_size */
  int32_t vt;           /* text vocab (lm_head rows) */
  int32_t max_rows;     /* capacity in rows of the trunk (prefill chunk / decode batch) */
  int32_t max_batch;    /* capacity in rows of the heads (lm_head + local decoder), <= max_rows */
  ua2_gpt_desc und, backbone, gen, decoder;
  const void* wte;      /* [vt, C] */
  const void* audio_emb;/* [va*n_cb, C] */
  const void* lm_head;  /* packed [vt, C] */
  const void* projection; /* packed [Cd, C] */
  const void* const* audio_head; /* n_cb packed [va, Cd] */
  /* sequence state (device) */
  int32_t* tokens;      /* [max_rows, n_cb+1] current input frame(s) */
  uint8_t* mask;        /* [max_rows, n_cb+1] */
  int32_t* row_pos;     /* [max_rows] */
  int32_t* row_seq;     /* [max_rows] */
  int32_t* dec_pos;     /* [n_cb, max_rows]: row i filled with i — positions of the local decoder's step i */
  int32_t* dec_seq;     /* [max_rows] = 0,1,2,...: page-table rows of the local decoder's 8-slot caches */
  int32_t* forbid;      /* [max_rows] forbid_prefix per row */
  int32_t* out_tokens;  /* [max_rows, n_cb+1] sampled [text, a0..a7] */
  int32_t* frame_log;   /* [log_frames, max_rows, n_cb+1] */
  int32_t* counters;    /* [4]: [0] frame index (log slot), [1] sampling draw index, [2..3] sampler seed (lo, hi) */
  int32_t log_frames;
  /* scratch (device, fp32 unless noted) — sizes in ua2_stage3_scratch_floats() */
  float* scratch;
  size_t scratch_floats;
} ua2_stage3_desc;

typedef struct ua2_stage3 ua2_stage3;

/* [round 6] Per-plan tables of the depth decoder (model_new.py:630-641).  `self.projection(_embed_audio(i, sample))` and layer 0's
 * q | k | v of that row at position i + 1 are functions of the sampled id alone; ua2_stage3_create builds them for all (n_cb - 1) * va
 * ids by running the frame's own launches on the null stream (a row's bits do not depend on the rows beside it), synchronises, and the
 * frame's arg-max gathers the rows instead of launching the two GEMVs per step (greedy frames of row-invariant plans; top-k sampling and
 * the order-free opt-in keep the launches).  Bit-identical ids and logits (tests/test_gpu_lm.py); the tables live in `scratch`:
 * ua2_stage3_scratch_floats() includes (n_cb - 1) * va * (Cd fp32 + Cd bf16 + Cd / 16 fp32 + q fp32 + k + v) — 1.9 GB at the released
 * sizes.  Environment, read when a plan is sized and created: UA2_NO_PROJ_TABLE=1 (neither table), UA2_NO_QKV_TABLE=1 (projection only). */
size_t ua2_stage3_scratch_floats(const ua2_stage3_desc* d);
int ua2_stage3_create(const ua2_stage3_desc* d, ua2_stage3** out);
void ua2_stage3_destroy(ua2_stage3* h);
/* Sampling mode of the heads: topk == 1 -> greedy (default); topk > 1 -> ua2_sample_topk with this temperature / seed.
 * The seed is written to counters[2..3] on `stream` (stream-ordered with the frames that follow); topk / temperature
 * select the captured graph, the seed does not. */
int ua2_stage3_set_sampling(ua2_stage3* h, int32_t topk, float temperature, uint64_t seed, void* stream);
/* Row groups (ua2_attn_args.group_*; device tables owned by the caller, alive until replaced) for the attention of the
 * NEXT ua2_stage3_trunk call (they are cleared when it returns): prefill rows of one sequence then share LDS-staged K/V
 * pages on the MFMA flash kernel.  ua2_stage3_frame (decode) never uses groups. */
int ua2_stage3_set_prefill_groups(ua2_stage3* h, const int32_t* group_rows, const int32_t* group_seq, const int32_t* group_nkeys,
                                  int32_t n_groups, int32_t group_q_tiles);
/* rows > 0 (UA2_BF16 plans; default 0 = off): launches of the four GPTs (trunk and depth decoder) with at least `rows` rows — prefill chunks of
 * batches (BASELINE config 3: 32 x 195 rows), decode frames of >= `rows` sequences — set ua2_linear_args.sum_order =
 * UA2_SUM_ORDER_FREE: the 256-row-tile GEMM with one chain over K (csrc/ua2_gemm2.hip), 10-25 % faster per launch at >= 2048
 * rows.  The price is the row-invariance contract ACROSS that threshold: a sequence prefilled alone (few rows) and inside a big
 * batch (>= rows) then differs by fp32 summation noise in its K/V cache, i.e. by bf16-level noise in later logits — the same
 * class of difference the MFMA prefill attention already has against decode rows.  Below the threshold nothing changes. */
int ua2_stage3_set_order_free_rows(ua2_stage3* h, int32_t rows);
/* cfg_scale > 1: frames of (conditional, unconditional) row pairs — rows 2p, 2p + 1; an even row count — sample from the guided
 * logits (ua2_cfg_mix); feedback mode 2 continues every row from its pair's conditional row. */
int ua2_stage3_set_cfg(ua2_stage3* h, float cfg_scale);

/* model_new.py:594-613 (embed-merge -> U-expert -> backbone -> G-expert -> blend) for R rows
 * described by tokens/mask/row_pos/row_seq.  Used for prefill (forward_prefix, :456-497; the
 * discarded lm_head/local-decoder work :498-506 is skipped) and as the first half of a frame. */
int ua2_stage3_trunk(ua2_stage3* h, int32_t R, void* stream);
/* model_new.py:617-641: lm_head + greedy text sample, then the 8-step local decoder.  [r6] On bf16 plans of up to 6 rows lm_head has no
 * launch of its own: its column tiles ride on workgroups past the grid of the local decoder's down-projection launches (which fill
 * half of the CUs; csrc/ua2_gemv.hip gemv_rider_kernel) and the text sample is taken behind the decoder — the same logits and ids
 * bit for bit (lm_head reads only the trunk's output, :617 vs :629-640).  UA2_NO_RIDER=1: lm_head as a launch, as before. */
int ua2_stage3_heads(ua2_stage3* h, int32_t R, void* stream);
/* Feedback for the next frame, on device (evaluation/tts_task.py:259-280 mode 0 "audio";
 * evaluation/asr_task.py:668-682 mode 1 "text"; mode 2 = "audio" with every row continuing from row 0's
 * sample, the classifier-free-guidance pair of tts_task.py:256-258,278-280): logs out_tokens, builds the next input frame,
 * row_pos += 1, applies the reason_eos -> forbid_prefix switch (tts_task.py:263-266). */
int ua2_stage3_feedback(ua2_stage3* h, int32_t R, int32_t mode, int32_t reason_eos, int32_t reason_card, void* stream);
/* trunk + heads + feedback (mode < 0: no feedback), captured once into a hipGraph and replayed
 * (use_graph != 0).
 * mode | UA2_FRAME_SKIP_TEXT_HEAD (audio-feedback modes 0 and 2 only): the frame skips lm_head and the text sample — in those
 * loops the text id is fed back under a zero mask and collected into a list the generators never read
 * (evaluation/tts_task.py:259,274-277; model_new.py:617 computes it every frame), so the audio ids are bit-identical with and
 * without it; the frame log's text column holds -1 for such frames and the next frame's (masked) text token is 0. */
#define UA2_FRAME_SKIP_TEXT_HEAD 16
/* mode 1 | UA2_FRAME_SKIP_AUDIO_EXPERTS [round 6]: every row of the frame is a TEXT step of a text-only continuation (the loops of
 * evaluation/asr_task.py:666-682 and its twins from their second frame on: the fed-back masks are (audio 0, text 1) and no audio step ever
 * follows).  There audio_step_mask = 0 multiplies both experts' outputs (model_new.py:607 backbone_input, :613 h_final) and nothing reads
 * their caches again, so the frame does not run audio_understanding_expert / audio_generation_expert (5 of the trunk's 33 layers at the
 * released sizes): text ids bit-identical (tests/test_gpu_lm.py).  The CALLER vouches for the precondition — the first frame after a prefill
 * (it consumes the prompt's last token, usually an audio step) and any sequence that will see an audio step later must run without it:
 * the skipped positions of the experts' caches are left unwritten. */
#define UA2_FRAME_SKIP_AUDIO_EXPERTS 32
int ua2_stage3_frame(ua2_stage3* h, int32_t R, int32_t mode, int32_t reason_eos, int32_t reason_card,
                     int32_t use_graph, void* stream);
/* Expose intermediate buffers for tests: name in {"h_final","text_logits","audio_logits"}. */
float* ua2_stage3_buffer(ua2_stage3* h, const char* name);

#ifdef __cplusplus
}
#endif
#endif /* UA2HIP_H */
