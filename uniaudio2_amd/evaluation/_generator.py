"""Task generators of the decode loop, host side: prompt layout, prefill, frame loop, EOS / phase logic.

Mirror of the reference's evaluation/*_task.py `Generator` classes (tts_task.py:53-285 and its
audiogen / musicgen / songen twins, which differ only in the special-token key; asr_task.py
prepare_asr_task :299-326, generate_asr :630-688; audio_understanding.py get_condition_seq :233-282,
generate_answer :284-339).  Same constructor, attributes, method names, argument meaning and return
values: (8, T) int32 reason / semantic tensors with the semantic offset removed and the first frame of
each phase dropped (tts_task.py:271,283-284), or the decoded text.

What differs is where the loop runs: the reference calls `generate_frame` from Python and
synchronises twice per frame to test EOS (tts_task.py:261,263); here frames run on the device in
chunks (`Model_stage3.generate_frames`: hipGraph replay, feedback and the reason_eos ->
forbid_prefix switch on device) and the host reads the id log once per chunk.  Frames computed past
the EOS frame are discarded, so the returned ids are those of the reference's loop.
"""
from typing import Tuple

import os

import torch

SPECIAL_TOKENS = {'<think>': 128002, '</think>': 128003, '</answer>': 128005,
                  '<transcription>': 128011, '</transcription>': 128012, '<lyric>': 128013,
                  '</lyric>': 128014, '<caption>': 128015, '</caption>': 128016, '<answer>': 128017,
                  '<reason_token>': 128018, '<semantic_token>': 128019}
TEXT_EOS = 128001            # evaluation/asr_task.py:674


class PhaseSplitter:
    """Host-side bookkeeping of the generation loop (tts_task.py:253-284), one frame at a time:
    stop at the semantic EOS frame, switch from the reason phase to the semantic phase at the
    reason EOS frame (which is not stored), strip the semantic offset, drop the first stored frame of
    each phase when stacking."""

    def __init__(self, reason_eos, semantic_eos, reason_card):
        self.reason_eos, self.sem_eos, self.reason_card = reason_eos, semantic_eos + reason_card, reason_card
        self.is_reason, self.save_flag, self.done = True, True, False
        self.pre_reason, self.pre_sem = [], []

    def push(self, audio):
        """audio: (1, 8) int tensor of one frame (conditional row).  Returns False once generation is over."""
        if self.done:
            return False
        if torch.all(audio == self.sem_eos):
            self.done = True
            return False
        if torch.all(audio == self.reason_eos):
            self.is_reason, self.save_flag = False, False
        if self.save_flag:
            if self.is_reason:
                self.pre_reason.append(audio)
            else:
                self.pre_sem.append(audio - self.reason_card)
        else:
            self.save_flag = True
        return True

    def result(self):
        if len(self.pre_reason) < 2 or len(self.pre_sem) < 2:
            from ..parallel import GenerationFailed
            raise GenerationFailed("stack expects a non-empty TensorList: the model produced no reason/semantic frames "
                                   "(same failure as the reference's torch.stack at tts_task.py:283-284)")
        de_reason = torch.stack(self.pre_reason[1:]).permute(1, 2, 0).squeeze(0)      # (8, T_r)
        de_sem = torch.stack(self.pre_sem[1:]).permute(1, 2, 0).squeeze(0)
        return de_reason, de_sem


class GeneratorBase:
    chunk_frames = 16        # frames per device-side chunk between host EOS checks
    # The audio-feedback loops never read the text id a frame samples (tts_task.py:259 appends it to a list nobody uses, :274-277
    # feed it back under text_mask = False): the on-device loop skips lm_head + its sample there (ua2hip.h
    # UA2_FRAME_SKIP_TEXT_HEAD; identical (reason, semantic)).  UA2_KEEP_TEXT_HEAD=1 computes it as the reference does.
    skip_text_head = os.environ.get("UA2_KEEP_TEXT_HEAD") is None

    def __init__(self, model, train_args, audio_tokenizer_config=None, audio_model_path=None,
                 text_tokenizer_path=None, is_cfg=False):
        from ..tools.tokenizer.Text2ID.text_tokenizer import load_text_tokenizer
        self._model = model
        self.is_cfg = is_cfg
        self._model.setup_caches(2 if is_cfg else 1)          # tts_task.py:64-67
        self._text_tokenizer = load_text_tokenizer(text_tokenizer_path)
        # the reference also loads the audio codec here (tts_task.py:71); generation never calls it,
        # so it is created lazily by the CLI stages that need it
        self._audio_tokenizer = None
        self.device = next(model.parameters()).device
        self.sample_rate = 24000
        self.empty_token = 0
        for k in ("text_pad_token", "semantic_pad_token", "semantic_eos", "semantic_bos", "reason_eos", "reason_bos",
                  "reason_pad_token", "parallel_number", "audio_reason_card"):
            setattr(self, k, getattr(train_args, k))
        self.audio_prompt_bos = getattr(train_args, "audio_prompt_bos", None)
        self.audio_prompt_eos = getattr(train_args, "audio_prompt_eos", None)
        self.special_token_dict = self.get_special_token()

    def get_special_token(self):
        return dict(SPECIAL_TOKENS)

    # ---- prompt layout ((T, 9) frames: 8 audio streams + 1 text stream, with masks) ------------
    def text_pad(self, x):
        seq = torch.zeros((len(x), self.parallel_number), dtype=torch.int64)
        seq[:, -1] = x
        seq[:, :-1] = self.empty_token
        return seq

    def audio_pad(self, x):
        seq = torch.full((x.shape[0], self.parallel_number), self.empty_token, dtype=torch.int64)
        seq[:, :-1] = x
        return seq

    def add_offset_semantic(self, x, offset_value):
        return x + offset_value

    def add_special_token(self, key, this_data):
        if key.startswith('text_seq'):
            return this_data
        key = key.replace('_seq', '')
        bos = torch.ones(1) * self.special_token_dict['<' + key + '>']
        eos = torch.ones(1) * self.special_token_dict['</' + key + '>']
        return torch.cat([bos, this_data, eos], dim=0)

    def reason_seq_bos_eos(self, d):
        bos = torch.ones(1, d.shape[1]) * self.reason_bos
        eos = torch.ones(1, d.shape[1]) * self.reason_eos
        return torch.cat([bos, d, eos], dim=0)

    def semantic_seq_bos_eos(self, d):
        bos = torch.ones(1, d.shape[1]) * self.semantic_bos
        eos = torch.ones(1, d.shape[1]) * self.semantic_eos
        return self.add_offset_semantic(torch.cat([bos, d, eos], dim=0), self.audio_reason_card)

    def _text_block(self, data):
        data = self.text_pad(data)
        mask = torch.zeros((data.shape[0], self.parallel_number))
        mask[:, -1] = True
        return data, mask

    def _audio_block(self, data):
        data = self.audio_pad(data)
        mask = torch.zeros((data.shape[0], self.parallel_number))
        mask[:, :-1] = True
        return data, mask

    def _prepare_text_conditioned(self, task_prompt, text_seq, key, cfg=False):
        """prepare_tts_task / prepare_tts_task_for_cfg and twins (tts_task.py:175-205)."""
        text_seq = self.add_special_token(key, text_seq)
        if cfg:
            task_prompt = torch.ones_like(task_prompt) * self.text_pad_token
            text_seq = torch.ones_like(text_seq) * self.text_pad_token
        pd, pm = self._text_block(task_prompt)
        td, tm = self._text_block(text_seq)
        return torch.cat([pd, td], dim=0), torch.cat([pm, tm], dim=0)

    def prepare_asr_task(self, task_prompt, this_reason_data, this_semantic_data):
        """asr_task.py:299-326: prompt text frames, then [reason_bos, reason.., reason_eos] and
        [sem_bos, sem.., sem_eos] (+ reason_card) audio frames."""
        td, tm = self._text_block(task_prompt)
        audio = torch.cat([self.reason_seq_bos_eos(this_reason_data), self.semantic_seq_bos_eos(this_semantic_data)], dim=0)
        ad, am = self._audio_block(audio)
        return torch.cat([td, ad], dim=0), torch.cat([tm, am], dim=0)

    def get_condition_seq(self, d, keys, types, task_prompt_data):
        """audio_understanding.py:233-282."""
        sequence, mask = [], []
        data, m = self._text_block(task_prompt_data)
        sequence.append(data); mask.append(m)
        for key, tp in zip(keys, types):
            if tp == 'text':
                data, m = self._text_block(self.add_special_token(key, d[key]))
            else:
                this = d[key].transpose(0, 1).long()
                if tp == 'audio_prompt':
                    this = self.semantic_seq_bos_eos(this)
                    bos = torch.ones(1, this.shape[1]) * self.audio_prompt_bos
                    eos = torch.ones(1, this.shape[1]) * self.audio_prompt_eos
                    this = torch.cat([bos, this[1:-1, :], eos], dim=0)
                elif key.startswith('reason_seq'):
                    this = self.reason_seq_bos_eos(this)
                else:
                    this = self.semantic_seq_bos_eos(this)
                data, m = self._audio_block(this)
            sequence.append(data); mask.append(m)
        return torch.cat(sequence, dim=0).to(torch.int64), torch.cat(mask, dim=0)

    # ---- loops ----------------------------------------------------------------------------------
    def _set_sampling(self, topk, temperature):
        self._model.set_sampling(topk, temperature)

    def _prefill(self, rows_tokens, rows_mask):
        """rows_*: list of (L, 9) prompts of equal length (1, or 2 with CFG; tts_task.py:228-245)."""
        dev = self.device
        tok = torch.stack(rows_tokens).to(dev)
        msk = torch.stack(rows_mask).bool().to(dev)
        B, L, _ = tok.shape
        pos = torch.arange(0, L, device=dev).unsqueeze(0).repeat(B, 1)
        self._model.reset_caches()
        self._model.forward_prefix(tok[:, :-1], labels=tok[:, 1:, :-1], tokens_mask=msk, loss_mask=msk,
                                   input_pos=pos[:, :-1])
        self._model.begin_decode(tok[:, -1:], msk[:, -1:], torch.tensor([L - 1], device=dev))
        return B, L

    @torch.inference_mode()
    def _generate_audio_tokens(self, tokens, tokens_mask, cfg_tokens=None, cfg_mask=None, topk=1, temperature=1.0,
                               max_audio_frames=500) -> Tuple[torch.Tensor, torch.Tensor]:
        """The loop of generate_tts / generate_audio / generate_LTS (tts_task.py:246-285)."""
        rows_t, rows_m = [tokens], [tokens_mask]
        if self.is_cfg:
            rows_t.append(cfg_tokens); rows_m.append(cfg_mask)
        B, L = self._prefill(rows_t, rows_m)
        self._set_sampling(topk, temperature)
        ph = PhaseSplitter(self.reason_eos, self.semantic_eos, self.audio_reason_card)
        frame = 0
        while not ph.done and frame < max_audio_frames:
            n = min(self.chunk_frames, max_audio_frames - frame)
            log = self._model.generate_frames(n, B, 2 if self.is_cfg else 0, reason_eos=self.reason_eos,
                                              reason_card=self.audio_reason_card, max_pos=L + max_audio_frames,
                                              skip_text_head=self.skip_text_head).cpu()       # (n, B, 9)
            for f in range(n):
                if not ph.push(log[f, 0:1, 1:]):                      # conditional row only (tts_task.py:256-258)
                    break
            frame += n
        de_reason, de_sem = ph.result()
        return de_reason.to(self.device), de_sem.to(self.device)

    @torch.inference_mode()
    def _generate_audio_tokens_batch(self, prompts, topk=1, temperature=1.0, max_audio_frames=500, cfg_prompts=None):
        """Many utterances at once on one GPU (the reference loops over them one by one,
        multi_task_inference.py:510): prompts = [(tokens (L_b, 9), mask (L_b, 9)), ...] of any lengths.  One ragged prefill,
        then all sequences decode together in device-side chunks; after each chunk the host runs every sequence's own
        phase / EOS bookkeeping (the same PhaseSplitter as the single-utterance loop) and retires the finished ones, so
        the batch shrinks as utterances end.  Greedy results equal the one-by-one results bit for bit (row invariance).
        With `is_cfg` (the reference's --use_cfg: an unconditional twin of every prompt, tts_task.py:228-245) utterance b
        occupies the row PAIR (2b, 2b + 1) = (conditional, unconditional): guidance mixes inside a pair
        (model_new.py:618-622,634-637), both rows continue from the pair's conditional sample (tts_task.py:278-280), the
        bookkeeping reads the conditional row, and a finished utterance retires both of its rows."""
        per = 2 if self.is_cfg else 1
        if self.is_cfg:
            if cfg_prompts is None or len(cfg_prompts) != len(prompts):
                raise ValueError("is_cfg: one unconditional prompt per prompt (prepare_*_for_cfg) is needed")
            for (t, _), (ct, _) in zip(prompts, cfg_prompts):
                if t.shape != ct.shape:
                    raise ValueError("a prompt and its unconditional twin must have the same shape")
            prompts = [p for pair in zip(prompts, cfg_prompts) for p in pair]
        B = len(prompts)                                             # rows
        st = getattr(self._model, "_st", None)
        longest = max(int(t.shape[0]) for t, _ in prompts)
        need_rows = sum(int(t.shape[0]) - 1 for t, _ in prompts)
        if st is None or st["B"] < B or st["max_rows"] < min(need_rows, 8192) or st["log_frames"] < max_audio_frames:
            self._model.setup_caches(B, max_rows=max(64, min(need_rows, 8192)), log_frames=max(512, max_audio_frames))
        self._model.begin_ragged([(t, m.bool()) for t, m in prompts])
        self._set_sampling(topk, temperature)
        n_utt = B // per
        splitters = [PhaseSplitter(self.reason_eos, self.semantic_eos, self.audio_reason_card) for _ in range(n_utt)]
        active, frame = list(range(n_utt)), 0                        # live utterances, in row(-pair) order
        while active and frame < max_audio_frames:
            n = min(self.chunk_frames, max_audio_frames - frame)
            log = self._model.generate_frames(n, per * len(active), 2 if self.is_cfg else 0, reason_eos=self.reason_eos,
                                              reason_card=self.audio_reason_card, max_pos=longest + max_audio_frames,
                                              skip_text_head=self.skip_text_head).cpu()       # (n, rows, 9)
            keep = []
            for r, b in enumerate(active):
                row = per * r                                        # the utterance's (conditional) row
                for f in range(n):
                    if not splitters[b].push(log[f, row:row + 1, 1:]):
                        break
                if not splitters[b].done:
                    keep.append(r)
            self._model.retire_rows([per * r + j for r in keep for j in range(per)], per * len(active))
            active = [active[r] for r in keep]
            frame += n
        out = []
        for ph in splitters:
            de_reason, de_sem = ph.result()
            out.append((de_reason.to(self.device), de_sem.to(self.device)))
        return out

    @torch.inference_mode()
    def _generate_text(self, tokens, tokens_mask, topk=1, temperature=1.0, max_frames=500) -> str:
        """The loop of generate_asr / generate_audio_caption / generate_answer (asr_task.py:658-688)."""
        B, L = self._prefill([tokens], [tokens_mask])
        self._set_sampling(topk, temperature)
        text, frame, done = [], 0, False
        while not done and frame < max_frames:
            n = min(self.chunk_frames, max_frames - frame)
            # text-only continuation: from its second frame on the understanding / generation experts are dead code (their outputs are
            # masked out and their caches never read again: UA2_FRAME_SKIP_AUDIO_EXPERTS, identical text ids; UA2_KEEP_AUDIO_EXPERTS=1 runs them)
            log = self._model.generate_frames(n, 1, 1, max_pos=L + max_frames,
                                              skip_audio_experts=os.environ.get("UA2_KEEP_AUDIO_EXPERTS") is None).cpu()
            for f in range(n):
                t = int(log[f, 0, 0])
                if t == TEXT_EOS:
                    done = True
                    break
                text.append(t)
            frame += n
        return self._text_tokenizer.decode(torch.tensor(text, dtype=torch.long))
