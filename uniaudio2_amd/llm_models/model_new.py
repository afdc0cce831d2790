"""Audio-text LM of the decode loop, host side.

Mirror of the reference's llm_models/model_new.py `Model_stage3` (:334-687) for its inference
methods: `setup_caches` (:554-565), `reset_caches` (:647-651), `forward_prefix` (:456-507),
`generate_frame` (:568-645) keep their signatures, the state-dict keys are identical
(`backbone.*`, `decoder.*`, `audio_understanding_expert.*`, `audio_generation_expert.*`,
`audio_embeddings.weight`, `projection.weight`, `audio_head`), so a reference checkpoint loads
with `load_state_dict`.

All arithmetic runs in libua2hip.so through the frame executor (include/ua2hip.h
ua2_stage3_*): one C call per prefill chunk / per frame, hipGraph replay for the frame.
`generate_frames` is the MI355X-native fast path the generators use: N frames back to back
with the sample -> next-input feedback, forbid_prefix switch and frame log all on device.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import Stage3Desc, check, lib, vp
from .config import Config as gpt_config
from .lit_model import GPT


@dataclass
class ModelArgs:
    llm_name: str
    decoder_name: str
    llm_pretrained_model: str
    audio_embeddings_path: str
    audio_understanding_expert_path: str
    audio_semantic_vocab_size: int
    audio_reason_vocab_size: int
    audio_num_codebooks: int


def ragged_schedule(n_frames):
    """Continuous-batching plan for fixed per-sequence frame counts: yields (step, active, keep) — run `step` frames with
    the sequences `active` (original indices, in row order), then keep the rows at positions `keep` (in order) as rows
    0..len(keep)-1.  Every sequence runs exactly n_frames[b] frames; a sequence leaves the batch the frame it finishes."""
    active = list(range(len(n_frames)))
    t_now = 0
    while active:
        step = min(int(n_frames[b]) for b in active) - t_now
        t_now += step
        keep = [r for r, b in enumerate(active) if int(n_frames[b]) > t_now]
        yield step, list(active), keep
        active = [active[r] for r in keep]


class Model_stage3(nn.Module):
    """Stage 3: text-audio joint model (inference)."""

    def __init__(self, config: ModelArgs, device=None):
        super().__init__()
        self.config = config
        self.backbone = GPT(gpt_config.from_name(config.llm_name), device=device)
        backbone_dim = self.backbone.config.n_embd
        self.decoder = GPT(gpt_config.from_name(config.decoder_name), device=device, with_embeddings=False)
        decoder_dim = self.decoder.config.n_embd
        va = config.audio_semantic_vocab_size + config.audio_reason_vocab_size
        self.audio_embeddings = nn.Embedding(va * config.audio_num_codebooks, backbone_dim, device=device)
        self.projection = nn.Linear(backbone_dim, decoder_dim, bias=False, device=device)
        self.audio_head = nn.Parameter(torch.empty(config.audio_num_codebooks, decoder_dim, va, device=device))
        self.audio_understanding_expert = GPT(gpt_config.from_name("meta-llama/Llama-3.2-Understanding"),
                                              device=device, with_embeddings=False)
        self.audio_generation_expert = GPT(gpt_config.from_name("meta-llama/Llama-3.2-Generation"),
                                           device=device, with_embeddings=False)
        self._h = None
        self._st = None
        self.sampling_seed = None      # set by callers that key the sampler themselves (CLI: per-utterance keys)
        self.order_free_rows = 0       # > 0: every plan setup_caches builds opts its many-row launches into the order-free GEMM (CLI --order_free_rows)

    # ---- caches / device plan ----------------------------------------------------------------
    def setup_caches(self, max_batch_size: int, dtype: Optional[torch.dtype] = None, max_seq_length: int = 2048,
                     max_rows: Optional[int] = None, log_frames: int = 512):
        """model_new.py:554-565: 2048-slot caches for the three 3072-d GPTs, `audio_num_codebooks`
        slots for the local decoder.  `dtype` selects the kernel precision (default: the
        parameters' dtype, as in the reference where `.to(dtype)` decides)."""
        p0 = self.projection.weight
        device, dtype = p0.device, (dtype or p0.dtype)
        if device.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback); call model.to('cuda')")
        cfg = self.config
        ncb, va = cfg.audio_num_codebooks, cfg.audio_semantic_vocab_size + cfg.audio_reason_vocab_size
        B = max_batch_size
        max_rows = max(max_rows or 64, B)
        self._destroy()
        for g in (self.audio_understanding_expert, self.backbone, self.audio_generation_expert):
            g.set_kv_cache(B, max_seq_length=max_seq_length, device=device, dtype=dtype)
        self.decoder.set_kv_cache(B, max_seq_length=ncb, device=device, dtype=dtype)
        st = dict(dtype=dtype, device=device, B=B, max_rows=max_rows, log_frames=log_frames, ncb=ncb, va=va)
        cast = lambda t: t.detach().to(device=device, dtype=dtype).contiguous()
        st["wte"] = cast(self.backbone.transformer.wte.weight)
        st["audio_emb"] = cast(self.audio_embeddings.weight)
        st["lm_head"] = ops.pack_linear(self.backbone.lm_head.weight.detach(), dtype)
        st["projection"] = ops.pack_linear(self.projection.weight.detach(), dtype)
        st["audio_head"] = [ops.pack_linear(self.audio_head[i].detach(), dtype, transposed=True) for i in range(ncb)]
        i32 = dict(dtype=torch.int32, device=device)
        st["tokens"] = torch.zeros(max_rows, ncb + 1, **i32)
        st["mask"] = torch.zeros(max_rows, ncb + 1, dtype=torch.uint8, device=device)
        st["row_pos"] = torch.zeros(max_rows, **i32)
        st["row_seq"] = torch.zeros(max_rows, **i32)
        st["dec_pos"] = torch.arange(ncb, **i32).unsqueeze(1).expand(ncb, max_rows).contiguous()
        st["dec_seq"] = torch.arange(max_rows, **i32)
        st["forbid"] = torch.zeros(max_rows, **i32)
        st["out_tokens"] = torch.zeros(max_rows, ncb + 1, **i32)
        st["frame_log"] = torch.zeros(log_frames, max_rows, ncb + 1, **i32)
        st["counters"] = torch.zeros(4, **i32)

        d = Stage3Desc()
        d.dtype, d.n_cb, d.va, d.vt = ops.dtype_code(dtype), ncb, va, self.backbone.config.padded_vocab_size
        d.max_rows, d.max_batch = max_rows, B
        gd = [self.audio_understanding_expert.desc(), self.backbone.desc(), self.audio_generation_expert.desc(),
              self.decoder.desc()]
        d.und, d.backbone, d.gen, d.decoder = gd
        d.wte, d.audio_emb = st["wte"].data_ptr(), st["audio_emb"].data_ptr()
        d.lm_head, d.projection = st["lm_head"].data_ptr(), st["projection"].data_ptr()
        ah = (vp * ncb)(*[t.data_ptr() for t in st["audio_head"]])
        d.audio_head = ah
        for k in ("tokens", "mask", "row_pos", "row_seq", "dec_pos", "dec_seq", "forbid", "out_tokens", "frame_log",
                  "counters"):
            setattr(d, k, st[k].data_ptr())
        d.log_frames = log_frames
        n = lib.ua2_stage3_scratch_floats(C.byref(d))
        st["scratch"] = torch.empty(n, dtype=torch.float32, device=device)
        d.scratch, d.scratch_floats = st["scratch"].data_ptr(), n
        h = vp()
        check(lib.ua2_stage3_create(C.byref(d), C.byref(h)), "ua2_stage3_create")
        st["keep"] = (d, gd, ah)
        self._h, self._st = h, st
        self._sampling = None
        self._cfg = 1.0
        self._pos_hi = 0
        self._text_fed_back = False    # the last frame run was a text-feedback frame (every row now holds masks audio 0 / text 1)
        if getattr(self, "order_free_rows", 0) > 0 and dtype == torch.bfloat16:
            self.set_order_free_rows(self.order_free_rows)

    def _destroy(self):
        if self._h is not None:
            lib.ua2_stage3_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def reset_caches(self):
        """model_new.py:647-651."""
        self._need()
        st = self._st
        st["counters"][0:1].zero_()      # frame-log slot; the sampler's draw index [1] keeps counting (like a generator)
        st["forbid"].zero_()

    def _need(self):
        if self._h is None:
            raise TypeError("You need to call `model.setup_caches()`")

    def _check_positions(self, max_pos: int):
        """Every position a launch touches must lie inside the caches planned by setup_caches: the QKV epilogue
        indexes the RoPE tables and the page table by position, so a position past the end would write K/V into
        another sequence's pages.  The reference fails loudly in the same situation (`index_copy_` on the
        2048-slot cache, lit_model.py:831-856); so does this."""
        limit = self.backbone.plan["max_seq"]
        if max_pos >= limit or max_pos < 0:
            raise ValueError(f"position {max_pos} is outside the KV cache / RoPE tables planned for max_seq_length="
                             f"{limit} (setup_caches); shorten the prompt, generate fewer frames or plan a longer cache")

    def _set_groups(self, pos, seq):
        """Row groups of the next prefill chunk for the MFMA flash attention (bf16 plans; the exact-fp32 plan keeps the
        row-by-row kernel).  pos, seq: the chunk's row positions / sequences (any device)."""
        st = self._st
        if st["dtype"] != torch.bfloat16:
            return
        cfg = self.backbone.config
        g = ops.attn_groups(pos.cpu().numpy(), seq.cpu().numpy(), cfg.n_head, cfg.n_query_groups, st["device"])
        st["groups"] = g                   # keeps the device tables alive while the executor points at them
        check(lib.ua2_stage3_set_prefill_groups(self._h, g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[0].shape[0], g[3]),
              "ua2_stage3_set_prefill_groups")

    def _load_rows(self, tokens, tokens_mask, pos, seq):
        """tokens (R, 9) any int dtype, mask (R, 9) bool, pos (R,), seq (R,) -> device state."""
        st = self._st
        R = tokens.shape[0]
        self._text_fed_back = False                # tokens / masks from outside: nothing is known about the rows' step kinds
        st["tokens"][:R].copy_(tokens)
        st["mask"][:R].copy_(tokens_mask)
        st["row_pos"][:R].copy_(pos)
        st["row_seq"][:R].copy_(seq)
        return R

    # ---- reference-compatible methods ---------------------------------------------------------
    @torch.inference_mode()
    def forward_prefix(self, tokens: torch.Tensor, labels: torch.Tensor = None, tokens_mask: torch.Tensor = None,
                       loss_mask: torch.Tensor = None, input_pos=None, input_pos_maxp1=None):
        """model_new.py:456-507 — fills the KV caches of the three trunk GPTs for `tokens`
        (B, S, 9) at `input_pos` (B, S).  `tokens_mask` is (B, S+1, 9) as the generators pass it
        (evaluation/tts_task.py:244); only [:, :-1] is used (:476-480).  The reference's lm_head /
        local-decoder pass over the prefix (:498-506) produces values every caller discards and is
        skipped; returns None."""
        self._need()
        st = self._st
        B, S, W = tokens.shape
        if input_pos is None:
            input_pos = torch.arange(S, device=tokens.device).unsqueeze(0).expand(B, S)
        mask = tokens_mask[:, :S]
        seq = torch.arange(B, device=tokens.device).unsqueeze(1).expand(B, S)
        tk, mk = tokens.reshape(B * S, W), mask.reshape(B * S, W)
        ps, sq = input_pos.reshape(-1), seq.reshape(-1)
        # time-major chunks so every chunk only needs KV of earlier chunks
        order = torch.argsort(ps, stable=True)
        tk, mk, ps, sq = tk[order], mk[order], ps[order], sq[order]
        self._check_positions(int(ps.max().item()))
        self._check_positions(int(ps.min().item()))
        R, mr = B * S, st["max_rows"]
        for s0 in range(0, R, mr):
            n = self._load_rows(tk[s0:s0 + mr], mk[s0:s0 + mr], ps[s0:s0 + mr], sq[s0:s0 + mr])
            self._set_groups(ps[s0:s0 + mr], sq[s0:s0 + mr])
            check(lib.ua2_stage3_trunk(self._h, n, ops.stream()), "ua2_stage3_trunk")
        return None

    @torch.inference_mode()
    def generate_frame(self, tokens: torch.Tensor, tokens_mask: torch.Tensor, input_pos: torch.Tensor,
                       input_pos_maxp1=None, temperature: float = 1.0, topk: int = 1, forbid_prefix: int = 0,
                       cfg_scale: float = 1.0) -> torch.Tensor:
        """model_new.py:568-645.  tokens (B, 1, 9), tokens_mask (B, 1, 9), input_pos (1,) shared
        or (B,) per sequence.  Returns (B, 9) int32 [text, a0..a7] on device."""
        self._need()
        self.set_sampling(topk, temperature)
        self.set_cfg(cfg_scale if tokens.size(0) > 1 else 1.0)
        if temperature <= 0:
            raise ValueError("temperature must be > 0")
        st = self._st
        B, S, W = tokens.shape
        assert S == 1 and W == st["ncb"] + 1, "last stream must be text"
        pos = input_pos.reshape(-1)
        if pos.numel() == 1:
            pos = pos.expand(B)
        seq = torch.arange(B, device=tokens.device)
        self._load_rows(tokens.reshape(B, W), tokens_mask.reshape(B, W), pos, seq)
        st["forbid"][:B].fill_(int(forbid_prefix))
        self._check_positions(int(pos.max().item()))
        self._check_positions(int(pos.min().item()))
        check(lib.ua2_stage3_frame(self._h, B, -1, 0, 0, 1, ops.stream()), "ua2_stage3_frame")
        return st["out_tokens"][:B].clone()

    def set_sampling(self, topk: int = 1, temperature: float = 1.0, seed: Optional[int] = None):
        """topk == 1: greedy (masked arg-max, lowest index on ties).  topk > 1: model_new.py:146-187 on device
        (top-k threshold, exponential-race draw) with a counter-based generator keyed by `seed` (default: the
        torch seed, multi_task_inference.py:159) — reproducible, but not torch's random stream."""
        self._need()
        if temperature <= 0:
            raise ValueError("temperature must be > 0")
        va = self._st["va"]
        if topk <= 0 or topk > va:
            raise ValueError(f"topk must be in 1..{va}")
        if seed is None:
            seed = self.sampling_seed if self.sampling_seed is not None else torch.initial_seed()
        key = (int(topk), float(temperature), int(seed) & (2 ** 64 - 1))
        if getattr(self, "_sampling", None) != key:
            # a new key rewinds the draw index on the device; the same key keeps counting (like a global generator)
            check(lib.ua2_stage3_set_sampling(self._h, key[0], key[1], key[2], ops.stream()), "ua2_stage3_set_sampling")
            self._sampling = key

    def set_cfg(self, cfg_scale: float = 1.0):
        """Classifier-free guidance (model_new.py:618-622, 634-637): with cfg_scale > 1 a frame of two rows
        (conditional, unconditional) samples from l[1] + (l[0] - l[1]) * cfg_scale and both rows take that sample.
        Applies to generate_frame and generate_frames (mode 2 continues every row from row 0)."""
        self._need()
        cfg_scale = max(float(cfg_scale), 1.0)
        if getattr(self, "_cfg", 1.0) != cfg_scale:
            check(lib.ua2_stage3_set_cfg(self._h, cfg_scale), "ua2_stage3_set_cfg")
            self._cfg = cfg_scale

    def set_order_free_rows(self, rows: int = 0):
        """bf16 plans: trunk launches of >= `rows` rows (batched prefill, decode frames of that many sequences) run on the
        order-free 256-row-tile GEMM (include/ua2hip.h ua2_stage3_set_order_free_rows; 0 = off, the default: every row keeps
        the bits of its single-sequence run)."""
        self._need()
        check(lib.ua2_stage3_set_order_free_rows(self._h, int(rows)), "ua2_stage3_set_order_free_rows")

    # ---- MI355X-native fast path ---------------------------------------------------------------
    @torch.inference_mode()
    def generate_frames(self, n_frames: int, batch: int, mode: int, reason_eos: int = -1, reason_card: int = 0,
                        max_pos: Optional[int] = None, use_graph: bool = True, frame_events=None,
                        skip_text_head: bool = False, skip_audio_experts: bool = False) -> torch.Tensor:
        """Runs `n_frames` frames back to back from the state left by the previous frame (first
        call: after `begin_decode`).  mode 0 = audio feedback (evaluation/tts_task.py:259-280),
        1 = text feedback (evaluation/asr_task.py:668-682; the depth decoder is skipped there — its samples are
        never fed back — so the audio columns of the log are zeros), 2 = guided pair.  Returns the log slice
        (n_frames, batch, 9) int32 (device).
        skip_text_head (modes 0 and 2): lm_head and the text sample are skipped — the audio loops feed the text id back under
        a zero mask and never read it (tts_task.py:259,274-277), so the audio columns are bit-identical with and without it;
        the text column of the log then holds -1.  The reference-shaped generate_frame always computes it."""
        if skip_text_head:
            if mode not in (0, 2):
                raise ValueError("skip_text_head applies to the audio-feedback modes (0, 2)")
            mode = mode | 16                                  # UA2_FRAME_SKIP_TEXT_HEAD
        if skip_audio_experts and mode != 1:
            raise ValueError("skip_audio_experts applies to the text-feedback mode (1)")
        self._need()
        st = self._st
        start = int(st["counters"][0].item())
        if start + n_frames > st["log_frames"]:
            raise ValueError("frame log too small: raise log_frames in setup_caches")
        # the last frame of this call reads / writes position (current max row_pos) + n_frames - 1.  The host-side bound
        # `_pos_hi` avoids a device sync; it can only over-estimate (after retirements), so the exact value is fetched
        # before refusing.  `max_pos` is accepted for compatibility with round-1 callers and ignored.
        limit = self.backbone.plan["max_seq"]
        if self._pos_hi + n_frames - 1 >= limit:
            self._pos_hi = int(st["row_pos"][:batch].max().item())
            self._check_positions(self._pos_hi + n_frames - 1)
        self._pos_hi += n_frames
        s = ops.stream()
        for i in range(n_frames):
            # skip_audio_experts (text-only continuations: asr_task.py:666-682 and twins): from the session's SECOND text-feedback frame on
            # every row is a text step fed back by the executor itself (masks audio 0 / text 1) — the first frame consumes the prompt's
            # last token, which may be an audio step, and runs whole.  UA2_FRAME_SKIP_AUDIO_EXPERTS = 32.
            m = mode | 32 if (skip_audio_experts and self._text_fed_back) else mode
            check(lib.ua2_stage3_frame(self._h, batch, m, reason_eos, reason_card, int(use_graph), s),
                  "ua2_stage3_frame")
            self._text_fed_back = mode == 1
            if frame_events is not None:           # measurement hook: one event after every frame (bench.py p50 / p99)
                frame_events[i].record()
        return st["frame_log"][start:start + n_frames, :batch]

    # ---- ragged batches / continuous batching (the reference has neither: SURVEY.md A.17, §8e) ------
    @torch.inference_mode()
    def forward_prefix_ragged(self, tokens_list, mask_list):
        """Prefill of prompts of different lengths in one pass.  tokens_list[b] (L_b, 9), mask_list[b] (L_b, 9):
        the rows to cache for sequence b (callers pass prompt[:-1] as with forward_prefix); sequence b takes
        page-table row b and positions 0..L_b-1.  Rows are issued time-major in chunks of `max_rows`, so a chunk
        only ever needs K/V written by itself or by earlier chunks."""
        self._need()
        st = self._st
        dev = st["device"]
        assert len(tokens_list) == len(mask_list) <= st["B"]
        tk = torch.cat([t.reshape(-1, t.shape[-1]) for t in tokens_list]).to(dev)
        mk = torch.cat([m.reshape(-1, m.shape[-1]) for m in mask_list]).to(dev)
        ps = torch.cat([torch.arange(t.shape[0], device=dev) for t in tokens_list])
        sq = torch.cat([torch.full((t.shape[0],), b, device=dev) for b, t in enumerate(tokens_list)])
        order = torch.argsort(ps, stable=True)
        tk, mk, ps, sq = tk[order], mk[order], ps[order], sq[order]
        self._check_positions(int(ps.max().item()))
        R, mr = tk.shape[0], st["max_rows"]
        for s0 in range(0, R, mr):
            n = self._load_rows(tk[s0:s0 + mr], mk[s0:s0 + mr], ps[s0:s0 + mr], sq[s0:s0 + mr])
            self._set_groups(ps[s0:s0 + mr], sq[s0:s0 + mr])
            check(lib.ua2_stage3_trunk(self._h, n, ops.stream()), "ua2_stage3_trunk")
        return None

    @torch.inference_mode()
    def retire_rows(self, keep, batch: int):
        """Continuous batching: of the `batch` live sequences (rows 0..batch-1 of the decode state) keep those at
        indices `keep`, in that order, as rows 0..len(keep)-1.  The per-row decode state and the page-table rows
        of the three trunk caches are permuted together (a permutation, so no page is lost); the local decoder's
        8-slot cache is rewritten every frame and needs nothing.  A row's results do not depend on its index or
        on its neighbours (tests/test_gpu_invariance.py), so retiring never changes what the survivors generate."""
        self._need()
        st = self._st
        keep = [int(k) for k in keep]
        assert len(set(keep)) == len(keep) and all(0 <= k < batch for k in keep)
        gone = [r for r in range(batch) if r not in set(keep)]
        perm = torch.tensor(keep + gone, dtype=torch.long, device=st["device"])
        for k in ("tokens", "mask", "row_pos", "forbid", "out_tokens"):
            st[k][:batch] = st[k][:batch][perm]
        for g in (self.audio_understanding_expert, self.backbone, self.audio_generation_expert):
            pt = g.kv_cache.page_table
            pt[:batch] = pt[:batch][perm]
        return len(keep)

    @torch.inference_mode()
    def begin_ragged(self, prompts):
        """Start of a batched generation: prompts[b] = (tokens (L_b, 9), mask (L_b, 9)) of any lengths.  Resets the caches
        and the page tables (undoing earlier retirements), prefills every prompt[:-1] in one ragged pass and loads each
        last prompt frame as the first decode frame of row b.  Returns the (B,) tensor of decode start positions."""
        self._need()
        st = self._st
        dev = st["device"]
        B = len(prompts)
        assert B <= st["B"], f"setup_caches({st['B']}) is too small for {B} sequences"
        self.reset_caches()
        for g in (self.audio_understanding_expert, self.backbone, self.audio_generation_expert):
            kv = g.kv_cache
            kv.page_table.copy_(torch.arange(kv.page_table.numel(), dtype=torch.int32, device=dev).view_as(kv.page_table))
        self.forward_prefix_ragged([t[:-1] for t, _ in prompts], [m[:-1] for _, m in prompts])
        last_t = torch.stack([t[-1] for t, _ in prompts]).to(dev)
        last_m = torch.stack([m[-1] for _, m in prompts]).to(dev)
        pos = torch.tensor([t.shape[0] - 1 for t, _ in prompts], device=dev)
        self.begin_decode(last_t.unsqueeze(1), last_m.unsqueeze(1), pos)
        return pos

    @torch.inference_mode()
    def generate_ragged(self, prompts, n_frames, mode: int = 0, reason_eos: int = -1, reason_card: int = 0, skip_text_head: bool = False):
        """Batched fixed-length generation with continuous batching: prompts[b] = (tokens (L_b, 9), mask (L_b, 9)),
        n_frames[b] frames for sequence b (deterministic stop).  All sequences decode
        together; a sequence leaves the batch the frame it finishes.  Returns a list of (n_frames[b], 9) int32
        id tensors (device), each bit-identical to the sequence's own B = 1 run."""
        B = len(prompts)
        assert len(n_frames) == B
        pos = self.begin_ragged(prompts)
        st = self._st
        dev = st["device"]
        out = [[] for _ in range(B)]
        max_pos = max(int(p) + int(n) for p, n in zip(pos.tolist(), n_frames)) + 1
        for step, active, keep in ragged_schedule(n_frames):
            n_act = len(active)
            if step > 0:
                log = self.generate_frames(step, n_act, mode, reason_eos, reason_card, max_pos=max_pos, skip_text_head=skip_text_head).clone()
                for r, b in enumerate(active):
                    out[b].append(log[:, r])
            self.retire_rows(keep, n_act)
        return [torch.cat(o) if o else torch.zeros(0, st["ncb"] + 1, dtype=torch.int32, device=dev) for o in out]

    def begin_decode(self, tokens, tokens_mask, input_pos, forbid_prefix=0):
        """Loads the first decode frame (the last prompt frame, tts_task.py:253-255) into the device state."""
        self._need()
        B, W = tokens.shape[0], tokens.shape[-1]
        pos = input_pos.reshape(-1)
        if pos.numel() == 1:
            pos = pos.expand(B)
        lo, hi = int(pos.min().item()), int(pos.max().item())
        self._check_positions(lo)
        self._check_positions(hi)
        self._pos_hi = hi
        self._load_rows(tokens.reshape(B, W), tokens_mask.reshape(B, W), pos, torch.arange(B, device=tokens.device))
        self._st["forbid"][:B].fill_(int(forbid_prefix))

    def buffer(self, name: str, rows: int):
        """Intermediate buffers for tests: 'h_final' (rows, C), 'text_logits' (rows, Vt), 'audio_logits' (rows, 8, Va)."""
        self._need()
        st = self._st
        p = lib.ua2_stage3_buffer(self._h, name.encode())
        if not p:
            raise KeyError(name)
        Cb = self.backbone.config.n_embd
        shape = {"h_final": (rows, Cb), "h": (rows, Cb), "text_logits": (rows, self.backbone.config.padded_vocab_size),
                 "audio_logits": (rows, st["ncb"], st["va"])}[name]
        base = st["scratch"]
        off = (C.cast(p, C.c_void_p).value - base.data_ptr()) // 4
        n = 1
        for s_ in shape:
            n *= s_
        return base[off:off + n].view(shape)

    def get_fsdp_wrap_module_list(self) -> List[nn.Module]:
        return (list(self.backbone.transformer.h) + list(self.audio_understanding_expert.transformer.h) +
                list(self.audio_generation_expert.transformer.h))
