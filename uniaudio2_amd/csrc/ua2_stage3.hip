// Whole-frame executor for Model_stage3's inference methods (llm_models/model_new.py:456-507
// forward_prefix, :568-645 generate_frame) and the generators' per-frame feedback
// (evaluation/tts_task.py:259-280, evaluation/asr_task.py:668-682).
//
// One C call issues every kernel of a frame on the caller's stream; all per-frame state
// (input tokens, step masks, positions, forbid_prefix, sampled ids, the frame log) lives in
// device memory, so the launch sequence is identical from frame to frame and is captured once
// into a hipGraph and replayed: no host round trip between the 350-odd kernels of a frame and
// none between frames (the reference synchronises twice per frame, tts_task.py:261,263).
#include <map>
#include <stdlib.h>
#include <string.h>
#include <tuple>
#include <vector>

#include "ua2_common.h"

struct ua2_stage3 {
  ua2_stage3_desc d;
  // deep copies of the per-layer pointer arrays
  std::vector<const void*> ptrs[4][5];
  std::vector<const float*> norms[4][2];
  std::vector<void*> pools[4][2];
  std::vector<const void*> audio_head;
  // scratch carve
  float *xa, *text, *xb, *hbuf, *xg, *hfin, *q, *act, *yattn, *xd, *curr_h;
  float *text_logits, *audio_logits, *pmax_t, *pmax_a;
  int32_t *pidx_t, *pidx_a;
  void* gemm_ws;               // operand scratch of the large-M linear kernel (max_rows x widest K)
  size_t gemm_ws_bytes;
  void* act_ws;                // packed SwiGLU output handed straight to the down-projection (max_rows x widest intermediate)
  // scaled-norm hand-over of the residual stream (UA2_PRO_SCALED, bf16): every producer of x also writes RNE_bf16(x (.) w_next)
  // for the RMSNorm + Linear that follows (row-major for launches of one row tile, fragment order otherwise) and the
  // per-16-column sums of squares — no prep launch and no in-kernel statistics between a residual update and its consumer
  void *xh, *xpk;
  float* ssq;
  // Projected-embedding table of the depth decoder (round 6; model_new.py:631 self.projection(ci_embed), :640 ci_embed = _embed_audio(i, ci_sample)):
  // row i * va + id = what the projection launch of step i + 1 would write for sample id of codebook i — y (fp32), and under the scaled
  // plan RNE_bf16(y (.) norm_1 of the decoder's layer 0) and the per-16-column sums of squares.  Built at create time by the very launches
  // the frame would run; the frame's arg-max gathers the row instead of launching the GEMV (ua2_misc.hip argmax_gather_kernel).
  float* ptab_y = nullptr;
  void* ptab_h = nullptr;
  float* ptab_ssq = nullptr;
  bool ptab_ho = false;
  // ... and layer 0's q | k | v of the depth decoder for the same rows (step i + 1 runs at position i + 1 whatever the frame: RoPE included):
  // q fp32 [qn], k / v [n_kv * head_size] of the plan dtype — the frame's arg-max writes q into the q buffer and k / v into the caches,
  // and layer 0's q|k|v launch of steps 1 .. n_cb - 1 disappears as well
  float* ptab_q = nullptr;
  void *ptab_k = nullptr, *ptab_v = nullptr;
  float* range_ws = nullptr;   // range-split scratch of the row-invariant 33-64-row down-projections (ua2_linear_args.range_ws): same bits
  size_t range_ws_bytes = 0;
  float* split_ws = nullptr;   // K-slab scratch of the order-free GEMM (ua2_linear_args.split_ws): handed to launches under UA2_SUM_ORDER_FREE only
  size_t split_ws_bytes = 0;
  bool scaled = false;
  int32_t npart_t, npart_a;
  int32_t topk = 1;            // 1 = greedy (fused arg-max partials); > 1 = ua2_sample_topk
  float temperature = 1.f;
  hipStream_t cap_stream = nullptr;                 // capture-only stream (the caller's may be the null stream, which cannot capture)
  float cfg_scale = 1.f;       // > 1: classifier-free guidance over a (conditional, unconditional) row pair
  int32_t order_free_rows = 0; // > 0 (bf16 plans): GPT launches (trunk and depth decoder) of at least this many rows take UA2_SUM_ORDER_FREE (ua2_stage3_set_order_free_rows)
  // row groups of the next ua2_stage3_trunk call (prefill): the trunk's attention then runs the MFMA flash kernel
  const int32_t *group_rows = nullptr, *group_seq = nullptr, *group_nkeys = nullptr;
  int32_t n_groups = 0, group_q_tiles = 0;
  std::map<std::tuple<int, int, int, int, int, int, int>, hipGraphExec_t> graphs;
};

namespace {

size_t align4(size_t n) { return (n + 3) & ~(size_t)3; }

struct Carve {
  size_t xa, text, xb, hbuf, xg, hfin, q, act, yattn, xd, curr_h, text_logits, audio_logits, pmax_t, pidx_t,
      pmax_a, pidx_a, gemm_ws, gemm_ws_floats, act_ws, xh, xpk, ssq, split, split_floats, ptab_y, ptab_h, ptab_ssq, ptab_rows, ptab_q, ptab_k, ptab_v, ptab_qkv, range, range_floats, total;
};

Carve carve(const ua2_stage3_desc& d) {
  Carve c;
  const size_t R = d.max_rows;
  const ua2_gpt_desc* gs[4] = {&d.und, &d.backbone, &d.gen, &d.decoder};
  size_t C = d.backbone.n_embd, Cd = d.decoder.n_embd, qmax = 0, actmax = 0;
  for (auto g : gs) {
    qmax = std::max(qmax, (size_t)g->n_head * g->head_size);
    actmax = std::max(actmax, (size_t)g->inter);
  }
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align4(n); return o; };
  c.xa = take(R * C); c.text = take(R * C); c.xb = take(R * C); c.hbuf = take(R * C); c.xg = take(R * C);
  c.hfin = take(R * C); c.q = take(R * qmax); c.act = take(R * actmax); c.yattn = take(R * qmax);
  c.xd = take(d.max_batch * Cd); c.curr_h = take(d.max_batch * C);
  const size_t npt = (d.vt + 15) / 16, npa = (d.va + 15) / 16, Bm = d.max_batch;
  c.text_logits = take(Bm * d.vt); c.audio_logits = take(Bm * d.n_cb * d.va);
  c.pmax_t = take(Bm * npt); c.pidx_t = take(Bm * npt); c.pmax_a = take(Bm * npa); c.pidx_a = take(Bm * npa);
  c.gemm_ws_floats = ua2_linear_workspace_bytes(d.dtype, (int64_t)R, (int64_t)std::max(std::max(C, Cd), std::max(qmax, actmax))) / sizeof(float);
  c.gemm_ws = take(c.gemm_ws_floats);
  c.act_ws = take(ua2_linear_workspace_bytes(d.dtype, (int64_t)R, (int64_t)actmax) / sizeof(float));
  const size_t Cw = std::max(C, Cd);
  c.xh = take(R * Cw / 2);                                       // bf16 rows
  c.xpk = take(ua2_linear_workspace_bytes(UA2_BF16, (int64_t)R, (int64_t)Cw) / sizeof(float));
  c.ssq = take(R * (Cw / 16 + 1));
  // K-slab scratch for order-free launches of a few hundred to a few thousand rows (tail tiles of a 300-tile grid; whole-grid slabs of
  // the narrow projections at ~1000 rows): plans that can hold such launches only
  c.split_floats = (d.dtype == UA2_BF16 && R >= 256) ? (size_t)16 << 20 : 0;
  c.split = take(c.split_floats);
  // range-split scratch: 16 ranges x 64 rows x the widest down-projection output, plans that can hold 33-64-row decode batches
  c.range_floats = (d.dtype == UA2_BF16 && d.max_batch >= 33 && getenv("UA2_RANGE_SPLIT") != nullptr) ? (size_t)16 * 64 * std::max(C, Cd) : 0;   // opt-in: measured slower in the frame (ua2_skinny.hip rsplit_*)
  c.range = take(c.range_floats);
  // projected-embedding table: (n_cb - 1) * va rows of [Cd fp32 | Cd bf16 | Cd / 16 fp32] (the last codebook's sample is never projected);
  // UA2_NO_PROJ_TABLE=1 keeps the per-step projection launch (A/B, and plans that cannot spare ~1.1 GB at the released sizes)
  c.ptab_rows = (d.n_cb > 1 && d.va > 0 && d.projection && d.audio_emb && Cd % 32 == 0 && getenv("UA2_NO_PROJ_TABLE") == nullptr) ? (size_t)(d.n_cb - 1) * d.va : 0;
  c.ptab_y = take(c.ptab_rows * Cd);
  c.ptab_h = take(c.ptab_rows * Cd / 2);
  c.ptab_ssq = take(c.ptab_rows * (Cd / 16));
  {   // q | k | v rows of the decoder's layer 0 (UA2_NO_QKV_TABLE=1: the projection table alone)
    const ua2_gpt_desc& g = d.decoder;
    const size_t esz = d.dtype == UA2_BF16 ? 2 : 4, kvw = (size_t)g.n_kv * g.head_size;
    c.ptab_qkv = (c.ptab_rows && g.n_layer > 0 && d.n_cb <= UA2_PAGE && (g.n_head * g.head_size) % 4 == 0 && (g.head_size * esz) % 16 == 0 &&
                  getenv("UA2_NO_QKV_TABLE") == nullptr) ? 1 : 0;
    const size_t rows = c.ptab_qkv ? c.ptab_rows : 0;
    c.ptab_q = take(rows * g.n_head * g.head_size);
    c.ptab_k = take((rows * kvw * esz + 3) / 4);
    c.ptab_v = take((rows * kvw * esz + 3) / 4);
  }
  c.total = off;
  return c;
}

// every ua2_linear of the executor may use the large-M kernel (same results, weights read once per 128 rows)
void fresh_args(const ua2_stage3* h, ua2_linear_args& a) {
  memset(&a, 0, sizeof(a));
  a.workspace = h->gemm_ws;
  a.workspace_bytes = h->gemm_ws_bytes;
}

// local = the depth decoder: positions < kLocalCtx, short-context attention (fused into the O-projection when R == 1)
// hand-over helpers: which form the consumer of a C-wide row reads (row-major: its launch is one row tile of the decode kernel)
struct Handover {
  ua2_stage3* h; int R, C; bool rows_h;
  Handover(ua2_stage3* h_, int R_, int C_) : h(h_), R(R_), C(C_), rows_h(R_ <= ua2_gemv_rows_preferred(h_->d.dtype, C_)) {}
  void consume(ua2_linear_args& a) const {
    a.prologue = UA2_PRO_SCALED; a.x_ssq = h->ssq; a.x = nullptr; a.norm_w = nullptr;
    if (rows_h) { a.x_h = h->xh; a.ldh = C; } else { a.x_packed = h->xpk; }
  }
  void produce(ua2_linear_args& a, const float* norm_w_next) const {
    if (!norm_w_next) return;
    a.y_norm_w = norm_w_next; a.y_ssq = h->ssq;
    if (rows_h) { a.y_h = h->xh; a.ldh = C; } else { a.y_packed = h->xpk; }
  }
  ua2_handover rowwise(const float* norm_w_next) const {
    ua2_handover ho{};
    ho.norm_w = norm_w_next; ho.ssq = h->ssq;
    if (rows_h) { ho.h = h->xh; ho.ldh = C; } else { ho.packed = h->xpk; }
    return ho;
  }
};

// launches of R rows under the plan's order-free opt-in (ua2_stage3_set_order_free_rows; bf16 plans)
bool order_free(const ua2_stage3* h, int R) { return h->d.dtype == UA2_BF16 && h->order_free_rows > 0 && R >= h->order_free_rows; }
// ... run the scaled hand-over whatever the plan's decode form (the order-free GEMM's epilogues carry it at any row count, and the
// two prep launches per layer go: profiles/r6_notes.md); the widths must be whole bf16 chunks as for h->scaled
bool order_free_scaled(const ua2_stage3* h, int R, bool prefill = false) {
  // Measured (profiles/r6_notes.md §3): the hand-over through ua2_gemm2.hip's epilogues does not pay in the frame — config-3 prefill
  // 46.43 (prep launches) vs 46.45 ms, B = 1024 decode 22.99 vs 24.55 ms per frame — and in prefill it doubles the distance to the
  // pinned plan (two different bf16 forms of the norm).  Default 0 = never; UA2_OF_SCALED = 1 decode frames, 2 prefill chunks too.
  static Ua2EnvInt mode{"UA2_OF_SCALED", 0};
  if (mode.get() < (prefill ? 2 : 1)) return false;
  return order_free(h, R) && h->d.backbone.n_embd % 32 == 0 && h->d.decoder.n_embd % 32 == 0 && getenv("UA2_NO_SCALED") == nullptr;
}

bool no_local_fuse() {
  static const bool v = getenv("UA2_NO_LOCAL_FUSE") != nullptr;   // A/B hook (profiles/r1_notes.md)
  return v;
}

// A one-row-tile GEMV (lm_head) whose column tiles travel on the idle CUs of a GPT's small launches (ua2_gemv.hip gemv_rider_kernel).
// Host kinds: 0 = q|k|v, 1 = o-projection, 2 = down-projection.  Tiles are dealt in launch order in proportion to a weight per
// kind, the last carrying launch takes whatever is left: every tile is computed exactly once per frame whatever the weights.
struct RiderPlan {
  ua2_linear_args r;
  int ntiles = 0, next = 0;
  double w[3] = {0, 0, 0}, wleft = 0;
  bool ok[3] = {false, false, false};
  void take(int kind, int& t0, int& t1) {
    const int left = ntiles - next;
    int n = (wleft <= w[kind] * 1.0001) ? left : (int)(left * (w[kind] / wleft) + 0.5);
    n = std::max(0, std::min(n, left));
    t0 = next; t1 = next + n; next = t1; wleft -= w[kind];
  }
};

// final_norm_w: weight of the RMSNorm + Linear that consumes this GPT's OUTPUT through UA2_PRO_SCALED (the depth decoder's
// ln_f in front of audio_head), or NULL (the trunk GPTs end in ua2_rmsnorm_blend, which reads the fp32 stream).
// With h->scaled the caller has already handed over x for layer 0 (embed_frame / rmsnorm_blend / the projection's epilogue).
// rp: the launches of this GPT that can (RiderPlan::ok) carry their share of the rider's column tiles.
int run_gpt(ua2_stage3* h, int gi, const ua2_gpt_desc& g, float* x, int R, const int32_t* row_pos,
            const int32_t* row_seq, hipStream_t s, bool local = false, bool grouped = false, const float* final_norm_w = nullptr,
            bool scaled = false, RiderPlan* rp = nullptr, bool qkv0_given = false) {
  auto launch = [&](const ua2_linear_args& a, int kind) -> int {
    if (rp && rp->ok[kind]) {
      int t0, t1;
      rp->take(kind, t0, t1);
      const int rc = ua2_gemv_launch_with_rider(a, rp->r, t0, t1, s);
      UA2_CHECK(rc <= 0, "ua2_stage3: launch kind %d cannot carry its rider (plan / launch mismatch)", kind);
      return rc;
    }
    return ua2_linear_launch(a, s);
  };
  const int dt = h->d.dtype;
  const int C = g.n_embd, qn = g.n_head * g.head_size, nqkv = (g.n_head + 2 * g.n_kv) * g.head_size;
  const Handover ho(h, R, C);
  // ua2_stage3_set_order_free_rows: many-row launches of the trunk on the 256-row-tile kernel (one chain over K); the launcher falls
  // back to the invariant kernels for anything outside that kernel's forms (scaled hand-over, small grids)
  const int order = order_free(h, R) ? UA2_SUM_ORDER_FREE : UA2_SUM_ORDER_INVARIANT;
  auto free_scratch = [&](ua2_linear_args& a) {           // K slabs are an order-free option: the invariant launches never see the scratch
    if (order == UA2_SUM_ORDER_FREE) { a.split_ws = h->split_ws; a.split_ws_bytes = h->split_ws_bytes; }
  };
  for (int l = 0; l < g.n_layer; ++l) {
    ua2_kv_geom kv{};                      // ring_pages = 0: the LM's caches are linear
    kv.k_pool = h->pools[gi][0][l]; kv.v_pool = h->pools[gi][1][l]; kv.page_table = g.page_table;
    kv.max_pages = g.max_pages; kv.n_kv = g.n_kv; kv.n_head = g.n_head; kv.head_size = g.head_size;

    ua2_linear_args a;
    fresh_args(h, a);
    a.dtype = dt; a.prologue = UA2_PRO_NORM; a.epilogue = UA2_EPI_QKV_ROPE;
    a.M = R; a.N = nqkv; a.K = C; a.x = x; a.ldx = C; a.norm_w = h->norms[gi][0][l]; a.eps = g.eps;
    a.w0 = h->ptrs[gi][0][l]; a.row_pos = row_pos; a.row_seq = row_seq; a.rope_cos = g.rope_cos;
    a.rope_sin = g.rope_sin; a.q_out = h->q; a.kv = kv;
    if (scaled) ho.consume(a);
    a.sum_order = order;
    if (!(qkv0_given && l == 0))        // depth decoder, steps >= 1: the previous step's arg-max wrote layer 0's q / k / v from the table
      if (int rc = launch(a, 0)) return rc;

    const bool fuse_attn = local && R == 1 && !no_local_fuse();
    // more than one row tile: the consumer runs a many-row kernel, so its producer writes the packed operand
    // directly and the consumer's prep launch disappears (same bits: the same RNE cast either way)
    static const bool no_handover = getenv("UA2_NO_PACKED_HANDOVER") != nullptr;   // A/B hook
    const int kc = dt == UA2_BF16 ? 32 : 16;
    const bool pack_o = !no_handover && R > ua2_gemv_rows_preferred(dt, qn) && qn % kc == 0;
    const bool pack_act = !no_handover && R > ua2_gemv_rows_preferred(dt, g.inter) && g.inter % kc == 0;
    if (!fuse_attn) {
      ua2_attn_args at;
      memset(&at, 0, sizeof(at));
      at.dtype = dt; at.R = R; at.q = h->q; at.row_pos = row_pos; at.row_seq = row_seq; at.kv = kv;
      if (pack_o) at.y_packed = h->gemm_ws; else at.y = h->yattn;
      if (grouped && !local && h->n_groups > 0) {
        at.group_rows = h->group_rows; at.group_seq = h->group_seq; at.group_nkeys = h->group_nkeys;
        at.n_groups = h->n_groups; at.group_q_tiles = h->group_q_tiles;
      }
      if (int rc = local ? ua2_attn_local_launch(at, s) : ua2_attn_launch(at, s)) return rc;
    }
    fresh_args(h, a);
    a.dtype = dt; a.prologue = fuse_attn ? UA2_PRO_LOCAL_ATTN : UA2_PRO_CAST; a.epilogue = UA2_EPI_RESIDUAL;
    a.M = R; a.N = C; a.K = qn; a.x = fuse_attn ? h->q : h->yattn; a.ldx = qn;
    a.w0 = h->ptrs[gi][1][l]; a.y = x; a.ldy = C; a.resid = x; a.ldr = C;
    if (fuse_attn) { a.row_pos = row_pos; a.row_seq = row_seq; a.kv = kv; }
    if (pack_o && !fuse_attn) a.x_packed = h->gemm_ws;
    if (scaled) ho.produce(a, h->norms[gi][1][l]);              // x after attention -> norm_2 + fc_1 / fc_2
    a.sum_order = order; free_scratch(a);
    if (int rc = launch(a, 1)) return rc;

    fresh_args(h, a);
    a.dtype = dt; a.prologue = UA2_PRO_NORM; a.epilogue = UA2_EPI_SWIGLU;
    a.M = R; a.N = g.inter; a.K = C; a.x = x; a.ldx = C; a.norm_w = h->norms[gi][1][l]; a.eps = g.eps;
    a.w0 = h->ptrs[gi][2][l]; a.w1 = h->ptrs[gi][3][l]; a.ldy = g.inter;
    if (pack_act) a.y_packed = h->act_ws; else a.y = h->act;
    if (scaled) ho.consume(a);
    a.sum_order = order;
    if (int rc = ua2_linear_launch(a, s)) return rc;

    fresh_args(h, a);
    a.dtype = dt; a.prologue = UA2_PRO_CAST; a.epilogue = UA2_EPI_RESIDUAL;
    a.M = R; a.N = C; a.K = g.inter; a.x = h->act; a.ldx = g.inter; a.w0 = h->ptrs[gi][4][l];
    a.y = x; a.ldy = C; a.resid = x; a.ldr = C;
    if (pack_act) a.x_packed = h->act_ws;
    if (scaled) ho.produce(a, l + 1 < g.n_layer ? h->norms[gi][0][l + 1] : final_norm_w);   // x after the MLP -> the next layer's norm_1 + qkv
    a.sum_order = order; free_scratch(a);
    a.range_ws = h->range_ws; a.range_ws_bytes = h->range_ws_bytes;      // 33-64 rows, K = 8192: one K range per workgroup + an in-order combine (same bits)
    if (int rc = launch(a, 2)) return rc;
  }
  return 0;
}

__global__ void bump_kernel(int32_t* c) { c[0] += 1; }

__global__ void feedback_kernel(int R, int ncb, int mode, int reason_eos, int reason_card, int log_frames,
                                int max_rows, int32_t* __restrict__ tokens, uint8_t* __restrict__ mask,
                                int32_t* __restrict__ row_pos, int32_t* __restrict__ forbid,
                                const int32_t* __restrict__ out, int32_t* __restrict__ log, int32_t* counters, int no_text) {
  const int frame = counters[0];
  const int w = ncb + 1;
  const bool audio_fb = (mode == 0 || mode == 2);
  for (int m = threadIdx.x; m < R; m += blockDim.x) {
    const int32_t* own = out + (size_t)m * w;
    if (frame < log_frames)
      for (int j = 0; j < w; ++j)   // text loop: no audio ids exist; UA2_FRAME_SKIP_TEXT_HEAD: no text id exists (logged as -1)
        log[((size_t)frame * max_rows + m) * w + j] = (j == 0 && no_text) ? -1 : ((audio_fb || j == 0) ? own[j] : 0);
    // mode 2 (classifier-free-guidance pairs, tts_task.py:256-258,278-280): every row continues from the sample of
    // its pair's conditional row (rows 2p, 2p + 1; the reference has the one pair)
    const int32_t* o = (mode == 2) ? out + (size_t)(m & ~1) * w : own;
    bool all_reason_eos = true;
    for (int i = 0; i < ncb; ++i) {
      all_reason_eos = all_reason_eos && (o[1 + i] == reason_eos);
      tokens[(size_t)m * w + i] = audio_fb ? o[1 + i] : 0;
      mask[(size_t)m * w + i] = audio_fb ? 1 : 0;
    }
    tokens[(size_t)m * w + ncb] = no_text ? 0 : o[0];           // fed back under a zero mask in the audio loop: any valid id
    mask[(size_t)m * w + ncb] = audio_fb ? 0 : 1;
    row_pos[m] += 1;
    if (audio_fb && all_reason_eos) forbid[m] = reason_card;  // tts_task.py:263-266
  }
  __syncthreads();
  if (threadIdx.x == 0) counters[0] = frame + 1;
}

// table rows row0 .. row0 + n - 1: position of the step that consumes row g = i * va + id is i + 1; page table of the scratch cache: row r -> page r
__global__ void ptab_pos_kernel(int32_t* __restrict__ out, int n, long long row0, int va) {
  const int r = threadIdx.x;
  if (r < 64) { out[r] = r < n ? (int)((row0 + r) / va) + 1 : 0; out[64 + r] = r; }
}

// rows [row0, row0 + n) of the embedding table (plan dtype) as fp32 rows: the operand of the table-building projection launches
template <int DT>
__global__ void emb_rows_f32_kernel(const void* __restrict__ emb, long long row0, int n, int C, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * C; i += (size_t)gridDim.x * blockDim.x)
    out[i] = load_elem<DT>(emb, (size_t)row0 * C + i);
}

}  // namespace

extern "C" size_t ua2_stage3_scratch_floats(const ua2_stage3_desc* d) { return d ? carve(*d).total : 0; }

extern "C" int ua2_stage3_create(const ua2_stage3_desc* d, ua2_stage3** out) {
  UA2_CHECK(d && out, "ua2_stage3_create: NULL argument");
  const Carve c = carve(*d);
  UA2_CHECK(d->scratch && d->scratch_floats >= c.total, "ua2_stage3_create: scratch too small (%zu < %zu floats)",
            d->scratch_floats, c.total);
  UA2_CHECK(d->max_batch > 0 && d->max_batch <= d->max_rows, "ua2_stage3_create: need 0 < max_batch <= max_rows");
  UA2_CHECK(d->und.n_embd == d->backbone.n_embd && d->gen.n_embd == d->backbone.n_embd,
            "ua2_stage3_create: expert width must equal backbone width (model_new.py:607,613)");
  ua2_stage3* h = new ua2_stage3();
  h->d = *d;
  const ua2_gpt_desc* gs[4] = {&d->und, &d->backbone, &d->gen, &d->decoder};
  ua2_gpt_desc* hs[4] = {&h->d.und, &h->d.backbone, &h->d.gen, &h->d.decoder};
  for (int gi = 0; gi < 4; ++gi) {
    const ua2_gpt_desc& g = *gs[gi];
    const void* const* src[5] = {g.qkv, g.proj, g.fc1, g.fc2, g.mlp_proj};
    for (int k = 0; k < 5; ++k) h->ptrs[gi][k].assign(src[k], src[k] + g.n_layer);
    h->norms[gi][0].assign(g.norm1, g.norm1 + g.n_layer);
    h->norms[gi][1].assign(g.norm2, g.norm2 + g.n_layer);
    h->pools[gi][0].assign(g.k_pool, g.k_pool + g.n_layer);
    h->pools[gi][1].assign(g.v_pool, g.v_pool + g.n_layer);
    hs[gi]->qkv = hs[gi]->proj = hs[gi]->fc1 = hs[gi]->fc2 = hs[gi]->mlp_proj = nullptr;  // use the copies
  }
  h->audio_head.assign(d->audio_head, d->audio_head + d->n_cb);
  float* b = d->scratch;
  h->xa = b + c.xa; h->text = b + c.text; h->xb = b + c.xb; h->hbuf = b + c.hbuf; h->xg = b + c.xg;
  h->hfin = b + c.hfin; h->q = b + c.q; h->act = b + c.act; h->yattn = b + c.yattn;
  h->xd = b + c.xd; h->curr_h = b + c.curr_h; h->text_logits = b + c.text_logits;
  h->audio_logits = b + c.audio_logits; h->pmax_t = b + c.pmax_t; h->pidx_t = (int32_t*)(b + c.pidx_t);
  h->pmax_a = b + c.pmax_a; h->pidx_a = (int32_t*)(b + c.pidx_a);
  h->gemm_ws = b + c.gemm_ws; h->gemm_ws_bytes = c.gemm_ws_floats * sizeof(float);
  h->act_ws = b + c.act_ws;
  h->xh = b + c.xh; h->xpk = b + c.xpk; h->ssq = b + c.ssq;
  h->split_ws = c.split_floats ? b + c.split : nullptr; h->split_ws_bytes = c.split_floats * sizeof(float);
  h->range_ws = c.range_floats ? b + c.range : nullptr; h->range_ws_bytes = c.range_floats * sizeof(float);
  // the scaled contract is the bf16 executor's (fp32 keeps the reference's operation order); A/B hook: UA2_NO_SCALED=1
  // Plans for more than 64 live sequences keep the prep form as well: at 256 rows the consumers' per-pass row-scale work and
  // the producers' fragment-order stores cost more than the 140 prep launches they replace (12.6 vs 13.7 ms per frame).  The
  // choice is made once per plan, so every row count served by one plan computes the same function (rows of a 64-sequence
  // batch are bit-identical to their B = 1 runs under the same plan).
  h->scaled = d->dtype == UA2_BF16 && getenv("UA2_NO_SCALED") == nullptr && d->backbone.n_embd % 32 == 0 && d->decoder.n_embd % 32 == 0 &&
              d->max_batch <= 64;
  h->npart_t = (d->vt + 15) / 16; h->npart_a = (d->va + 15) / 16;
  if (c.ptab_rows) {
    // The table is what the frame's own projection launches would write: same entry point, same arguments, rows in groups of up to 64
    // (decode kernel / weights-stationary kernel: a row's bits do not depend on its group).  Null stream, once per plan.
    h->ptab_y = b + c.ptab_y; h->ptab_h = b + c.ptab_h; h->ptab_ssq = b + c.ptab_ssq;
    h->ptab_ho = h->scaled;
    const int C = d->backbone.n_embd, Cd = d->decoder.n_embd;
    const int chunk = std::min(64, d->max_rows);          // operand rows staged in the trunk's first row buffer (max_rows x C): a B = 1 plan still builds 64 rows per launch
    void *tmp_k = nullptr, *tmp_v = nullptr;
    int32_t* tmp_i = nullptr;          // [64] positions, [64] page table (row r -> page r)
    if (c.ptab_qkv) {
      h->ptab_q = b + c.ptab_q; h->ptab_k = b + c.ptab_k; h->ptab_v = b + c.ptab_v;
      const size_t pool = (size_t)64 * d->decoder.n_kv * UA2_PAGE * d->decoder.head_size * (d->dtype == UA2_BF16 ? 2 : 4);
      if (hipMalloc(&tmp_k, pool) != hipSuccess || hipMalloc(&tmp_v, pool) != hipSuccess || hipMalloc((void**)&tmp_i, 128 * sizeof(int32_t)) != hipSuccess) {
        ua2_set_error("ua2_stage3_create: scratch cache of the q|k|v table"); delete h; return -1;
      }
    }
    for (size_t row0 = 0; row0 < c.ptab_rows; row0 += chunk) {
      const int n = (int)std::min<size_t>(chunk, c.ptab_rows - row0);
      if (d->dtype == UA2_BF16) hipLaunchKernelGGL(emb_rows_f32_kernel<UA2_BF16>, dim3(std::min(1024, n * 4)), dim3(256), 0, nullptr, d->audio_emb, (long long)row0, n, C, h->xa);
      else hipLaunchKernelGGL(emb_rows_f32_kernel<UA2_F32>, dim3(std::min(1024, n * 4)), dim3(256), 0, nullptr, d->audio_emb, (long long)row0, n, C, h->xa);
      ua2_linear_args a;
      fresh_args(h, a);
      a.dtype = d->dtype; a.prologue = UA2_PRO_CAST; a.epilogue = UA2_EPI_STORE;
      a.M = n; a.N = Cd; a.K = C; a.x = h->xa; a.ldx = C; a.w0 = d->projection; a.y = h->ptab_y + row0 * Cd; a.ldy = Cd;
      if (h->ptab_ho) {
        a.y_norm_w = h->norms[3][0][0]; a.y_ssq = h->ptab_ssq + row0 * (Cd / 16);
        a.y_h = reinterpret_cast<unsigned short*>(h->ptab_h) + row0 * Cd; a.ldh = Cd;
      }
      const Handover hod(h, n, Cd);
      const bool scaled_q = h->ptab_ho && c.ptab_qkv;
      if (scaled_q && !hod.rows_h) a.y_packed = h->xpk;        // the q|k|v launch below reads the fragment-order form, as in the frame
      if (int rc = ua2_linear_launch(a, nullptr)) { delete h; return rc; }
      if (!c.ptab_qkv) continue;
      // layer 0's q|k|v of these rows, exactly as run_gpt launches it: row r of the group = its own one-page sequence of a scratch cache
      const ua2_gpt_desc& g = d->decoder;
      const int qn = g.n_head * g.head_size, nqkv = (g.n_head + 2 * g.n_kv) * g.head_size, esz = d->dtype == UA2_BF16 ? 2 : 4;
      hipLaunchKernelGGL(ptab_pos_kernel, dim3(1), dim3(64), 0, nullptr, tmp_i, n, (long long)row0, d->va);
      ua2_kv_geom kv{};
      kv.k_pool = tmp_k; kv.v_pool = tmp_v; kv.page_table = tmp_i + 64; kv.max_pages = 1; kv.n_kv = g.n_kv; kv.n_head = g.n_head; kv.head_size = g.head_size;
      fresh_args(h, a);
      a.dtype = d->dtype; a.prologue = UA2_PRO_NORM; a.epilogue = UA2_EPI_QKV_ROPE;
      a.M = n; a.N = nqkv; a.K = Cd; a.x = h->ptab_y + row0 * Cd; a.ldx = Cd; a.norm_w = h->norms[3][0][0]; a.eps = g.eps;
      a.w0 = h->ptrs[3][0][0]; a.row_pos = tmp_i; a.row_seq = nullptr; a.rope_cos = g.rope_cos; a.rope_sin = g.rope_sin;
      a.q_out = h->ptab_q + row0 * qn; a.kv = kv;
      if (scaled_q) {
        a.prologue = UA2_PRO_SCALED; a.x_ssq = h->ptab_ssq + row0 * (Cd / 16); a.x = nullptr; a.norm_w = nullptr;
        if (hod.rows_h) { a.x_h = reinterpret_cast<unsigned short*>(h->ptab_h) + row0 * Cd; a.ldh = Cd; } else { a.x_packed = h->xpk; }
      }
      if (int rc = ua2_linear_launch(a, nullptr)) { delete h; return rc; }
      const size_t kvw = (size_t)g.n_kv * g.head_size * esz;
      if (int rc = ua2_kv_rows_extract(tmp_k, tmp_v, tmp_i, n, g.n_kv, g.head_size, esz, (char*)h->ptab_k + row0 * kvw, (char*)h->ptab_v + row0 * kvw, nullptr)) { delete h; return rc; }
    }
    const hipError_t e = hipStreamSynchronize(nullptr);
    if (tmp_k) { (void)hipFree(tmp_k); (void)hipFree(tmp_v); (void)hipFree(tmp_i); }
    if (e != hipSuccess) { ua2_set_error("ua2_stage3_create: building the projected-embedding table failed: %s", hipGetErrorString(e)); delete h; return -1; }
  }
  *out = h;
  return 0;
}

extern "C" void ua2_stage3_destroy(ua2_stage3* h) {
  if (!h) return;
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  delete h;
}

// identity: rows are sequences 0..R-1 (decode frames) -> no row_seq indirection in the kernels
// skip_experts (UA2_FRAME_SKIP_AUDIO_EXPERTS): every row of the frame is a TEXT step of a text-only continuation — the understanding and
// generation experts' outputs are multiplied by audio_step_mask = 0 (model_new.py:607, :613) and their caches are never read again, so
// the two GPTs are not run; the blends see zeros in their place and select the text / backbone operand exactly as before.
static int trunk_impl(ua2_stage3* h, int32_t R, bool identity, hipStream_t s, bool skip_experts = false) {
  UA2_CHECK(h && R > 0 && R <= h->d.max_rows, "ua2_stage3_trunk: R=%d out of range", R);
  ua2_stage3_desc d = h->d;
  if (identity) d.row_seq = nullptr;
  const int C = d.backbone.n_embd, w = d.n_cb + 1;
  // Decode frames run the scaled hand-over; PREFILL chunks (row groups set: the MFMA flash attention path) keep the
  // prep + RMSNorm-prologue form: prefill rows already differ from decode rows by bf16-level summation order,
  // and there the hand-over's epilogue costs more than the prep launches it saves (the tiled kernel's RESIDUAL epilogue
  // with the emission spills its accumulators: 8192-row prefill 82 -> 131 ms, profiles/r3_notes.md)
  const bool grouped = !identity;   // prefill chunks (ua2_stage3_trunk) may carry row groups; decode frames never do
  // (under the order-free opt-in the hand-over rides in ua2_gemm2.hip's epilogues at any row count: prefill chunks and plans for more
  // than 64 sequences take it too)
  const bool prefill = grouped && (h->n_groups > 0 || R > h->d.max_batch);   // ua2_stage3_trunk with row groups, or with more rows than sequences: a prefill chunk
  const bool scaled = (h->scaled && !prefill) || order_free_scaled(h, R, prefill);
  const Handover ho(h, R, C);
  ua2_handover e0{}, e1{}, e2{};
  if (scaled) { e0 = ho.rowwise(h->norms[0][0][0]); e1 = ho.rowwise(h->norms[1][0][0]); e2 = ho.rowwise(h->norms[2][0][0]); }
  if (int rc = ua2_embed_frame(d.dtype, R, C, d.n_cb, d.va, d.tokens, d.mask, d.audio_emb, d.wte, h->xa, h->text, &e0, s)) return rc;
  if (!skip_experts)
    if (int rc = run_gpt(h, 0, d.und, h->xa, R, d.row_pos, d.row_seq, s, false, grouped, nullptr, scaled)) return rc;
  // backbone_input = h_audio*audio_step + text_embeds*text_step   (model_new.py:607)
  if (int rc = ua2_rmsnorm_blend(R, C, h->xa, d.und.ln_f, d.und.eps, h->text, d.mask, w, 0, d.n_cb, h->xb, nullptr, &e1, s)) return rc;
  if (int rc = run_gpt(h, 1, d.backbone, h->xb, R, d.row_pos, d.row_seq, s, false, grouped, nullptr, scaled)) return rc;
  // h = ln_f(x); generation_input = h*audio_step                   (model_new.py:609-610)
  if (int rc = ua2_rmsnorm_blend(R, C, h->xb, d.backbone.ln_f, d.backbone.eps, nullptr, d.mask, w, 0, -1, h->xg, h->hbuf, &e2, s)) return rc;
  if (!skip_experts)
    if (int rc = run_gpt(h, 2, d.gen, h->xg, R, d.row_pos, d.row_seq, s, false, grouped, nullptr, scaled)) return rc;
  // h_final = h_audio*audio_step + h*text_step                     (model_new.py:613)
  if (int rc = ua2_rmsnorm_blend(R, C, h->xg, d.gen.ln_f, d.gen.eps, h->hbuf, d.mask, w, 0, d.n_cb, h->hfin, nullptr, nullptr, s)) return rc;
  return 0;
}

// (re-)seeding also rewinds the draw index (counters[1]): an utterance's samples are then a function of its own key and its
// own frame count, whatever was generated before it on this rank (per-utterance keys alone did not make the
// samples independent of the sharding — the index kept counting across utterances)
__global__ void set_seed_kernel(int32_t* c, uint32_t lo, uint32_t hi) { c[1] = 0; c[2] = (int32_t)lo; c[3] = (int32_t)hi; }

// The seed lives in device memory (counters[2..3]) and the samplers read it there, so a captured frame graph
// serves every seed: re-seeding between utterances is one 8-byte store on the caller's stream, never a re-capture.
extern "C" int ua2_stage3_set_sampling(ua2_stage3* h, int32_t topk, float temperature, uint64_t seed, void* stream) {
  UA2_CHECK(h != nullptr, "ua2_stage3_set_sampling: NULL handle");
  UA2_CHECK(temperature > 0.f, "temperature must be > 0");
  UA2_CHECK(topk >= 1 && topk <= h->d.va, "topk must be in 1..%d", h->d.va);
  h->topk = topk; h->temperature = temperature;
  hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, h->d.counters, (uint32_t)seed, (uint32_t)(seed >> 32));
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_stage3_set_prefill_groups(ua2_stage3* h, const int32_t* group_rows, const int32_t* group_seq, const int32_t* group_nkeys,
                                             int32_t n_groups, int32_t group_q_tiles) {
  UA2_CHECK(h != nullptr, "ua2_stage3_set_prefill_groups: NULL handle");
  UA2_CHECK(n_groups == 0 || (group_rows && group_seq && group_nkeys && group_q_tiles > 0), "ua2_stage3_set_prefill_groups: missing tables");
  h->group_rows = group_rows; h->group_seq = group_seq; h->group_nkeys = group_nkeys;
  h->n_groups = n_groups; h->group_q_tiles = group_q_tiles;
  return 0;
}

extern "C" int ua2_stage3_set_order_free_rows(ua2_stage3* h, int32_t rows) {
  UA2_CHECK(h != nullptr && rows >= 0, "ua2_stage3_set_order_free_rows: NULL handle or rows < 0");
  if (h->order_free_rows != rows) {                  // recorded frames bake the kernel choice: drop them
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
  }
  h->order_free_rows = rows;
  return 0;
}

extern "C" int ua2_stage3_set_cfg(ua2_stage3* h, float cfg_scale) {
  UA2_CHECK(h != nullptr && cfg_scale >= 1.f, "ua2_stage3_set_cfg: NULL handle or cfg_scale < 1");
  h->cfg_scale = cfg_scale;
  return 0;
}

extern "C" int ua2_stage3_trunk(ua2_stage3* h, int32_t R, void* stream) {
  UA2_CHECK(h != nullptr, "ua2_stage3_trunk: NULL handle");
  const int rc = trunk_impl(h, R, false, (hipStream_t)stream);
  h->n_groups = 0;                 // row groups describe ONE chunk: they never outlive the call they were set for
  return rc;
}

// text_only: the text head and its sample only.  The on-device text loop (feedback mode 1, asr_task.py:668-682) feeds
// back zeros for the audio streams, so what the depth decoder would sample is never read: skipping its 8 passes leaves
// the text ids unchanged ("waste removal with identical outputs") and removes ~45 % of the frame.
// skip_text: the depth decoder only.  In the audio-feedback loops (feedback modes 0 and 2 of the TTS / TTA / TTM / LTS / S2S
// generators) the sampled text id is fed back under a ZERO mask (evaluation/tts_task.py:274-277: text_mask = False) and appended
// to a list nobody reads (:259), so the frame's audio ids do not depend on it: skipping lm_head + its arg-max (788 MB of weights
// at 3072 x 128256 bf16 = 117 us of the 3.1 ms B = 1 frame, 220 us at 64 rows) leaves (reason, semantic) bit-identical
// (model_new.py:617 computes it every frame).  generate_frame's (B, 9) API never sets it.
static int heads_impl(ua2_stage3* h, int32_t R, bool text_only, void* stream, bool skip_text = false) {
  UA2_CHECK(h && R > 0 && R <= h->d.max_batch, "ua2_stage3_heads: R=%d out of range", R);
  hipStream_t s = (hipStream_t)stream;
  const ua2_stage3_desc& d = h->d;
  const int C = d.backbone.n_embd, Cd = d.decoder.n_embd, w = d.n_cb + 1;
  // the plan's order-free opt-in covers the heads and the projection as well (arg-max partials and the hand-over are forms of
  // ua2_gemm2.hip since round 6); `scaled` as in trunk_impl
  const int order = order_free(h, R) ? UA2_SUM_ORDER_FREE : UA2_SUM_ORDER_INVARIANT;
  const bool scaled = h->scaled || order_free_scaled(h, R);
  ua2_linear_args a;
  // text_logits = lm_head(last_h); greedy text sample             (model_new.py:617,623)
  fresh_args(h, a);
  a.dtype = d.dtype; a.prologue = UA2_PRO_CAST; a.epilogue = UA2_EPI_STORE;
  a.M = R; a.N = d.vt; a.K = C; a.x = h->hfin; a.ldx = C; a.w0 = d.lm_head; a.y = h->text_logits; a.ldy = d.vt;
  a.part_max = h->pmax_t; a.part_idx = h->pidx_t; a.sum_order = order;
  UA2_CHECK(!(skip_text && text_only), "ua2_stage3_heads: nothing left to compute");
  // lm_head does not feed the depth decoder (model_new.py:617 vs :629-640).  A forked graph branch for it replays 0.4 ms per frame
  // SLOWER than the linear chain on ROCm 7.2 (3.49 vs 3.09 ms, re-measured in round 6), so it stays in the chain — but not as a
  // launch of its own when the depth decoder's down-projections can carry it: each of their n_cb x n_layer launches fills half the
  // device (128 workgroups: the per-CU ingest cap, ua2_gemv.hip) and takes one slice of lm_head's column tiles on the other half.
  // Same bits as lm_head's own launch; the text sample moves behind the depth decoder (nothing in between reads it).
  const ua2_linear_args lm = a;
  RiderPlan rp;
  bool ride = false;
  if (!skip_text && !text_only && d.n_cb > 0 && d.decoder.n_layer > 0) {
    const ua2_gpt_desc& g = d.decoder;
    const int qn = g.n_head * g.head_size, nqkv = (g.n_head + 2 * g.n_kv) * g.head_size;
    ua2_linear_args hs[3];
    for (auto& v : hs) { fresh_args(h, v); v.dtype = d.dtype; v.M = R; v.x = h->act; }
    hs[0].prologue = scaled ? UA2_PRO_SCALED : UA2_PRO_NORM; hs[0].epilogue = UA2_EPI_QKV_ROPE; hs[0].N = nqkv; hs[0].K = Cd;
    hs[1].prologue = (R == 1 && d.n_cb <= 8 && !no_local_fuse()) ? UA2_PRO_LOCAL_ATTN : UA2_PRO_CAST; hs[1].epilogue = UA2_EPI_RESIDUAL; hs[1].N = Cd; hs[1].K = qn;
    hs[2].prologue = UA2_PRO_CAST; hs[2].epilogue = UA2_EPI_RESIDUAL; hs[2].N = Cd; hs[2].K = g.inter;
    const double wk[3] = {0.0, 0.0, 1.0};    // the down-projections only: riders on the 6-us q|k|v / o launches made the frame slower (profiles/r6_notes.md)
    rp.r = lm; rp.ntiles = (lm.N + 15) / 16;
#ifdef UA2_RIDER_EXPERIMENTS
    static Ua2EnvInt frac{"UA2_RIDER_TIMING_PCT", 100};   // timing only (wrong text ids): only this share of lm_head's tiles is computed at all
    rp.ntiles = (int)((long)rp.ntiles * frac.get() / 100);
#endif
    for (int k = 0; k < 3; ++k) {
      rp.ok[k] = wk[k] > 0 && ua2_gemv_rider_ok(hs[k], lm);
      rp.w[k] = rp.ok[k] ? wk[k] : 0.0;
      rp.wleft += rp.w[k] * d.n_cb * g.n_layer;
      ride = ride || rp.ok[k];
    }
  }
  if (!skip_text && !ride)
    if (int rc = ua2_linear_launch(lm, s)) return rc;
  // model_new.py:618-622: with guidance the sampler sees l[1] + (l[0] - l[1]) * cfg_scale and both rows take its sample
  const bool cfg = h->cfg_scale > 1.f && R > 1;
  UA2_CHECK(!cfg || R % 2 == 0, "ua2_stage3_heads: classifier-free guidance needs (conditional, unconditional) row pairs, R=%d", R);
  const int key_shift = cfg ? 1 : 0;                       // the two rows of a pair hold the same guided logits and draw the same numbers
  auto text_tail = [&]() -> int {
    if (skip_text) return 0;   // the sampler streams are keyed by (seed, draw index, row, stream id), so the audio streams' draws do not move
    if (cfg)
      if (int rc = ua2_cfg_mix(h->text_logits, d.vt, d.vt, h->cfg_scale, nullptr, h->pmax_t, h->pidx_t, R / 2, s)) return rc;
    if (h->topk == 1)
      return ua2_argmax_embed(d.dtype, R, h->npart_t, h->pmax_t, h->pidx_t, d.out_tokens, w, 0, nullptr, 0, C, nullptr, s);
    // model_new.py:623 sample_topk(text_logits, topk, temperature)
    return ua2_sample_topk(d.dtype, R, h->text_logits, d.vt, d.vt, std::min(h->topk, d.vt), h->temperature, nullptr, 0, d.counters + 1, 0,
                           d.out_tokens, w, 0, nullptr, 0, C, nullptr, key_shift, s);
  };
  if (!ride)
    if (int rc = text_tail()) return rc;
  const float* curr = h->hfin;
  // steps 1 .. n_cb - 1 take their projected input (and its hand-over) from the table, gathered by the previous step's arg-max
  // (under the order-free opt-in as well: the table rows are the row-invariant kernels' sums, one of the orders that contract allows)
  const bool tab = h->ptab_y && h->topk == 1 && (!scaled || h->ptab_ho || h->ptab_q);
  for (int i = 0; i < (text_only ? 0 : d.n_cb); ++i) {             // model_new.py:630-641
    const Handover hod(h, R, Cd);
    if (!(tab && i > 0)) {
      fresh_args(h, a);
      a.dtype = d.dtype; a.prologue = UA2_PRO_CAST; a.epilogue = UA2_EPI_STORE;
      a.M = R; a.N = Cd; a.K = C; a.x = curr; a.ldx = C; a.w0 = d.projection; a.y = h->xd; a.ldy = Cd;
      if (scaled) hod.produce(a, h->norms[3][0][0]);
      a.sum_order = order;
      if (int rc = ua2_linear_launch(a, s)) return rc;
    }
    if (int rc = run_gpt(h, 3, d.decoder, h->xd, R, d.dec_pos + (size_t)i * d.max_rows, nullptr, s, d.n_cb <= 8, false,
                         scaled ? d.decoder.ln_f : nullptr, scaled, ride ? &rp : nullptr, tab && i > 0 && h->ptab_q)) return rc;
    fresh_args(h, a);
    a.dtype = d.dtype; a.prologue = UA2_PRO_NORM; a.epilogue = UA2_EPI_STORE;
    a.M = R; a.N = d.va; a.K = Cd; a.x = h->xd; a.ldx = Cd; a.norm_w = d.decoder.ln_f; a.eps = d.decoder.eps;
    a.w0 = h->audio_head[i]; a.y = h->audio_logits + (size_t)i * d.va; a.ldy = d.n_cb * d.va;
    a.part_max = h->pmax_a; a.part_idx = h->pidx_a; a.forbid = d.forbid;
    if (scaled) hod.consume(a);
    a.sum_order = order;
    if (int rc = ua2_linear_launch(a, s)) return rc;
    if (cfg)   // model_new.py:634-637
      if (int rc = ua2_cfg_mix(h->audio_logits + (size_t)i * d.va, d.n_cb * d.va, d.va, h->cfg_scale, d.forbid, h->pmax_a,
                               h->pidx_a, R / 2, s)) return rc;
    if (tab && i + 1 < d.n_cb) {
      const ua2_handover g = hod.rowwise(nullptr);
      ua2_qkv_gather qg{};
      if (h->ptab_q) {
        const ua2_gpt_desc& gd = d.decoder;
        qg.tab_q = h->ptab_q; qg.tab_k = h->ptab_k; qg.tab_v = h->ptab_v; qg.q_out = h->q; qg.qn = gd.n_head * gd.head_size;
        qg.esz = d.dtype == UA2_BF16 ? 2 : 4; qg.pos = i + 1;
        qg.kv.k_pool = h->pools[3][0][0]; qg.kv.v_pool = h->pools[3][1][0]; qg.kv.page_table = gd.page_table; qg.kv.max_pages = gd.max_pages;
        qg.kv.n_kv = gd.n_kv; qg.kv.n_head = gd.n_head; qg.kv.head_size = gd.head_size;
      }
      // with layer 0's q | k | v in hand nothing reads the hand-over of the projected row (its only consumer was that launch)
      if (int rc = ua2_argmax_gather(R, h->npart_a, h->pmax_a, h->pidx_a, d.out_tokens, w, 1 + i, h->ptab_y, h->ptab_h, h->ptab_ssq,
                                     (int64_t)i * d.va, Cd, h->xd, (scaled && !h->ptab_q) ? &g : nullptr, h->ptab_q ? &qg : nullptr, s)) return rc;
    } else if (h->topk == 1) {
      if (int rc = ua2_argmax_embed(d.dtype, R, h->npart_a, h->pmax_a, h->pidx_a, d.out_tokens, w, 1 + i, d.audio_emb,
                                    i * d.va, C, h->curr_h, s)) return rc;
    } else {   // model_new.py:639 audio_sample_topk(ci_logits, topk, temperature, forbid_prefix)
      if (int rc = ua2_sample_topk(d.dtype, R, h->audio_logits + (size_t)i * d.va, d.n_cb * d.va, d.va, h->topk, h->temperature,
                                   d.forbid, 0, d.counters + 1, 1 + i, d.out_tokens, w, 1 + i, d.audio_emb, i * d.va, C, h->curr_h,
                                   key_shift, s)) return rc;
    }
    curr = h->curr_h;
  }
  if (ride) {
    UA2_CHECK(rp.next == rp.ntiles, "ua2_stage3_heads: %d of %d lm_head tiles rode", rp.next, rp.ntiles);
    if (int rc = text_tail()) return rc;
  }
  if (h->topk != 1) {   // one draw index per frame, advanced after every sampler of the frame has read it
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, d.counters + 1);
    UA2_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int ua2_stage3_heads(ua2_stage3* h, int32_t R, void* stream) { return heads_impl(h, R, false, stream); }

static int feedback_impl(ua2_stage3* h, int32_t R, int32_t mode, int32_t reason_eos, int32_t reason_card, void* stream, int no_text) {
  UA2_CHECK(h && R > 0 && R <= h->d.max_rows && mode >= 0 && mode <= 2, "ua2_stage3_feedback: bad arguments");
  const ua2_stage3_desc& d = h->d;
  hipLaunchKernelGGL(feedback_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, R, d.n_cb, mode, reason_eos,
                     reason_card, d.log_frames, d.max_rows, d.tokens, d.mask, d.row_pos, d.forbid, d.out_tokens,
                     d.frame_log, d.counters, no_text);
  UA2_LAUNCH_CHECK();
  return 0;
}

extern "C" int ua2_stage3_feedback(ua2_stage3* h, int32_t R, int32_t mode, int32_t reason_eos, int32_t reason_card,
                                   void* stream) {
  return feedback_impl(h, R, mode, reason_eos, reason_card, stream, 0);
}

extern "C" int ua2_stage3_frame(ua2_stage3* h, int32_t R, int32_t mode, int32_t reason_eos, int32_t reason_card,
                                int32_t use_graph, void* stream) {
  UA2_CHECK(h != nullptr, "ua2_stage3_frame: NULL handle");
  hipStream_t s = (hipStream_t)stream;
  const bool skip_text = mode >= 0 && (mode & UA2_FRAME_SKIP_TEXT_HEAD) != 0;
  const bool skip_experts = mode >= 0 && (mode & UA2_FRAME_SKIP_AUDIO_EXPERTS) != 0;
  if (mode >= 0) mode &= ~(UA2_FRAME_SKIP_TEXT_HEAD | UA2_FRAME_SKIP_AUDIO_EXPERTS);
  UA2_CHECK(!skip_text || mode == 0 || mode == 2, "ua2_stage3_frame: UA2_FRAME_SKIP_TEXT_HEAD goes with the audio-feedback modes (0, 2)");
  UA2_CHECK(!skip_experts || mode == 1, "ua2_stage3_frame: UA2_FRAME_SKIP_AUDIO_EXPERTS goes with the text-feedback mode (1)");
  auto body = [&](hipStream_t st) -> int {
    if (int rc = trunk_impl(h, R, true, st, skip_experts)) return rc;
    if (int rc = heads_impl(h, R, mode == 1, st, skip_text)) return rc;
    if (mode < 0) return 0;
    return feedback_impl(h, R, mode, reason_eos, reason_card, st, skip_text ? 1 : 0);
  };
  if (!use_graph) return body(s);
  int tbits, cbits;
  memcpy(&tbits, &h->temperature, sizeof(int));
  memcpy(&cbits, &h->cfg_scale, sizeof(int));
  const auto key = std::make_tuple((int)R, (int)mode | (skip_text ? UA2_FRAME_SKIP_TEXT_HEAD : 0) | (skip_experts ? UA2_FRAME_SKIP_AUDIO_EXPERTS : 0), (int)reason_eos, (int)reason_card, (int)h->topk, tbits, cbits);
  auto it = h->graphs.find(key);
  if (it == h->graphs.end()) {
    hipGraph_t graph = nullptr;
    if (!h->cap_stream) UA2_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    UA2_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = body(h->cap_stream);
    const hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
    if (rc) return rc;
    if (e != hipSuccess) {
      ua2_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
      return -2;
    }
    hipGraphExec_t exec = nullptr;
    UA2_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    it = h->graphs.emplace(key, exec).first;
  }
  UA2_HIP(hipGraphLaunch(it->second, s));
  return 0;
}

extern "C" float* ua2_stage3_buffer(ua2_stage3* h, const char* name) {
  if (!h || !name) return nullptr;
  if (!strcmp(name, "h_final")) return h->hfin;
  if (!strcmp(name, "text_logits")) return h->text_logits;
  if (!strcmp(name, "audio_logits")) return h->audio_logits;
  if (!strcmp(name, "h")) return h->hbuf;
  return nullptr;
}
