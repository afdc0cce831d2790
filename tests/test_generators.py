"""Host logic of the generators (no GPU): prompt layout and the phase / EOS bookkeeping, against a
direct restatement of the reference's loops (evaluation/tts_task.py:192-205,253-284; asr_task.py:299-326)."""
import types

import pytest
import torch

from uniaudio2_amd.evaluation._generator import GeneratorBase, PhaseSplitter

TA = types.SimpleNamespace(text_pad_token=128004, semantic_pad_token=9, semantic_eos=8193, semantic_bos=8192,
                           reason_eos=4097, reason_bos=4096, reason_pad_token=7, parallel_number=9,
                           audio_reason_card=4100)


def reference_loop(frames, reason_eos, semantic_eos, reason_card):
    """tts_task.py:253-284 restated on a given stream of (1, 8) samples."""
    is_reason, save_flag = True, True
    pre_r, pre_s = [], []
    for audio in frames:
        if torch.all(audio == (semantic_eos + reason_card)):
            break
        if torch.all(audio == reason_eos):
            is_reason = False
            save_flag = False
        if save_flag:
            if is_reason:
                pre_r.append(audio)
            else:
                pre_s.append(audio - reason_card)
        else:
            save_flag = True
    return (torch.stack(pre_r[1:]).permute(1, 2, 0).squeeze(0), torch.stack(pre_s[1:]).permute(1, 2, 0).squeeze(0))


@pytest.mark.parametrize("n_reason,n_sem,tail", [(5, 9, 3), (2, 2, 0), (20, 50, 7)])
def test_phase_splitter_equals_reference_loop(n_reason, n_sem, tail):
    g = torch.Generator().manual_seed(n_reason * 100 + n_sem)
    frames = [torch.randint(0, 4096, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_reason)]
    frames.append(torch.full((1, 8), TA.reason_eos, dtype=torch.int32))
    frames += [torch.randint(4100, 4100 + 8192, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_sem)]
    frames.append(torch.full((1, 8), TA.semantic_eos + TA.audio_reason_card, dtype=torch.int32))
    frames += [torch.randint(0, 100, (1, 8), generator=g, dtype=torch.int32) for _ in range(tail)]   # computed past EOS, discarded
    ph = PhaseSplitter(TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
    for f in frames:
        ph.push(f)
    r, s = ph.result()
    rr, rs = reference_loop(frames, TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
    assert torch.equal(r, rr) and torch.equal(s, rs)
    assert r.shape == (8, n_reason - 1) and s.shape == (8, n_sem - 1) and r.dtype == torch.int32


def test_phase_splitter_raises_like_reference_when_empty():
    ph = PhaseSplitter(TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
    ph.push(torch.zeros(1, 8, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ph.result()


def _gen():
    g = GeneratorBase.__new__(GeneratorBase)
    for k, v in vars(TA).items():
        setattr(g, k, v)
    g.empty_token = 0
    g.special_token_dict = g.get_special_token()
    return g


def test_tts_prompt_layout():
    g = _gen()
    prompt = torch.tensor([128000, 11, 12, 128001]); text = torch.tensor([128000, 21, 22, 23, 128001])
    data, mask = g._prepare_text_conditioned(prompt, text, 'transcription_seq')
    assert data.shape == (4 + 5 + 2, 9) and data.dtype == torch.int64
    assert data[:, :-1].eq(0).all() and mask[:, :-1].eq(0).all() and mask[:, -1].eq(1).all()
    assert data[4, -1] == 128011 and data[-1, -1] == 128012 and data[5:10, -1].tolist() == text.tolist()
    cd, cm = g._prepare_text_conditioned(prompt, text, 'transcription_seq', cfg=True)
    assert cd.shape == data.shape and cd[:, -1].eq(TA.text_pad_token).all()


def test_asr_prompt_layout():
    g = _gen()
    prompt = torch.tensor([128000, 5, 128001])
    reason = torch.arange(3 * 8).view(3, 8); sem = torch.arange(4 * 8).view(4, 8)
    data, mask = g.prepare_asr_task(prompt, reason, sem)
    assert data.shape == (3 + 5 + 6, 9)
    assert data[3, :8].eq(TA.reason_bos).all() and data[7, :8].eq(TA.reason_eos).all()
    assert data[8, :8].eq(TA.semantic_bos + TA.audio_reason_card).all()
    assert data[13, :8].eq(TA.semantic_eos + TA.audio_reason_card).all()
    assert torch.equal(data[9:13, :8], sem + TA.audio_reason_card)
    assert mask[:3, -1].eq(1).all() and mask[3:, :8].eq(1).all() and mask[3:, -1].eq(0).all()


class _ScriptedModel:
    """Stands in for Model_stage3 on CPU: replays a scripted id log through the generate_frames interface."""

    def __init__(self, log):
        self.log, self.cursor, self.calls = log, 0, []
        self._p = torch.nn.Parameter(torch.zeros(1))

    def parameters(self):
        return iter([self._p])

    def setup_caches(self, b): self.batch = b
    def reset_caches(self): self.cursor = 0
    def forward_prefix(self, *a, **k): pass
    def begin_decode(self, *a, **k): pass

    def set_sampling(self, topk, temperature, seed=None): self.sampling = (topk, temperature)

    def generate_frames(self, n, batch, mode, reason_eos=-1, reason_card=0, max_pos=None, skip_text_head=False, skip_audio_experts=False):
        self.calls.append((n, batch, mode))
        out = self.log[self.cursor:self.cursor + n]
        self.cursor += n
        return out


@pytest.mark.parametrize("n_reason,n_sem", [(5, 9), (16, 16), (15, 17), (30, 41)])
def test_generate_tts_chunked_loop_equals_reference_loop(n_reason, n_sem):
    """EOS inside / at the edge of a 16-frame device chunk; frames computed past EOS are discarded."""
    from uniaudio2_amd.evaluation.tts_task import Generator
    g = torch.Generator().manual_seed(n_reason)
    frames = [torch.randint(0, 4096, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_reason)]
    frames.append(torch.full((1, 8), TA.reason_eos, dtype=torch.int32))
    frames += [torch.randint(4100, 12292, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_sem)]
    frames.append(torch.full((1, 8), TA.semantic_eos + TA.audio_reason_card, dtype=torch.int32))
    frames += [torch.randint(0, 100, (1, 8), generator=g, dtype=torch.int32) for _ in range(40)]
    log = torch.zeros(len(frames), 1, 9, dtype=torch.int32)
    log[:, 0, 1:] = torch.cat(frames)
    gen = Generator(_ScriptedModel(log), TA, text_tokenizer_path="ids")
    r, s = gen.generate_tts(torch.tensor([128000, 1, 128001]), "tts", text_token=torch.tensor([128000, 2, 3, 128001]), topk=1)
    rr, rs = reference_loop(frames, TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
    assert torch.equal(r, rr) and torch.equal(s, rs) and r.dtype == torch.int32
    assert all(c[2] == 0 for c in gen._model.calls)
    gen._model.cursor = 0
    gen.generate_tts(torch.tensor([1]), "tts", text_token=torch.tensor([2]), topk=50, temperature=0.9)
    assert gen._model.sampling == (50, 0.9)                    # forwarded to the device sampler


def test_instruct_tts_prompt_layout_and_loop():
    """insturct_tts_task.py:201-219 (prompt + <caption> + <transcription> text frames), :171-198 (CFG twin = text_pad)."""
    from uniaudio2_amd.evaluation.insturct_tts_task import Generator
    g = Generator.__new__(Generator)
    for k, v in vars(TA).items():
        setattr(g, k, v)
    g.empty_token = 0
    g.special_token_dict = g.get_special_token()
    prompt, cap, text = torch.tensor([128000, 11, 128001]), torch.tensor([31, 32]), torch.tensor([21, 22, 23])
    data, mask = g.prepare_instruct_tts_task(prompt, cap, text)
    assert data.shape == (3 + 4 + 5, 9) and mask[:, -1].eq(1).all() and mask[:, :-1].eq(0).all()
    assert data[:, -1].tolist() == [128000, 11, 128001, 128015, 31, 32, 128016, 128011, 21, 22, 23, 128012]
    cd, cm = g.prepare_instruct_tts_task_for_cfg(prompt, cap, text)
    assert cd.shape == data.shape and cd[:, -1].eq(TA.text_pad_token).all() and torch.equal(cm, mask)
    # the loop: scripted frames, CFG pair -> mode 2, conditional row read
    gen_rng = torch.Generator().manual_seed(3)
    frames = [torch.randint(0, 4096, (1, 8), generator=gen_rng, dtype=torch.int32) for _ in range(4)]
    frames.append(torch.full((1, 8), TA.reason_eos, dtype=torch.int32))
    frames += [torch.randint(4100, 12292, (1, 8), generator=gen_rng, dtype=torch.int32) for _ in range(6)]
    frames.append(torch.full((1, 8), TA.semantic_eos + TA.audio_reason_card, dtype=torch.int32))
    frames += [torch.zeros(1, 8, dtype=torch.int32)] * 8
    log = torch.zeros(len(frames), 2, 9, dtype=torch.int32)
    log[:, 0, 1:] = torch.cat(frames)
    log[:, 1, 1:] = 77                                           # unconditional row: never read
    gen = Generator(_ScriptedModel(log), TA, text_tokenizer_path="ids", is_cfg=True)
    r, s = gen.generate_instruct_tts(prompt, "instruct_tts", text_token=text, caption=cap, topk=1)
    rr, rs = reference_loop(frames, TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
    assert torch.equal(r, rr) and torch.equal(s, rs)
    assert gen._model.batch == 2 and all(c[1] == 2 and c[2] == 2 for c in gen._model.calls)


def test_speech_s2s_condition_sequence_and_loop():
    """speech_s2s.py:233-281 + multi_task_inference.py:421-423: the source utterance's reason / semantic tokens with
    their bos / eos frames (semantic ids offset by the reason cardinality) after the prompt; never guided."""
    from uniaudio2_amd.evaluation.speech_s2s import Generator, S2S_KEYS, S2S_TYPES
    frames = [torch.randint(0, 4096, (1, 8), dtype=torch.int32) for _ in range(3)]
    frames.append(torch.full((1, 8), TA.reason_eos, dtype=torch.int32))
    frames += [torch.randint(4100, 12292, (1, 8), dtype=torch.int32) for _ in range(3)]
    frames.append(torch.full((1, 8), TA.semantic_eos + TA.audio_reason_card, dtype=torch.int32))
    frames += [torch.zeros(1, 8, dtype=torch.int32)] * 8
    log = torch.zeros(len(frames), 1, 9, dtype=torch.int32)
    log[:, 0, 1:] = torch.cat(frames)
    gen = Generator(_ScriptedModel(log), TA, text_tokenizer_path="ids", is_cfg=True)
    reason, sem = torch.arange(8 * 3).view(8, 3), torch.arange(8 * 5).view(8, 5)
    d = {"reason_seq_1": reason, "semantic_seq_1": sem, "reason_seq_2": reason, "semantic_seq_2": sem}
    prompt = torch.tensor([128000, 7, 128001])
    data, mask = gen.get_condition_seq(d, S2S_KEYS[:-2], S2S_TYPES[:-2], prompt)
    assert data.shape == (3 + 5 + 7, 9)
    assert data[3, :8].eq(TA.reason_bos).all() and data[7, :8].eq(TA.reason_eos).all()
    assert torch.equal(data[4:7, :8], reason.t()) and torch.equal(data[9:14, :8], sem.t() + TA.audio_reason_card)
    assert data[8, :8].eq(TA.semantic_bos + TA.audio_reason_card).all() and data[14, :8].eq(TA.semantic_eos + TA.audio_reason_card).all()
    r, s = gen.generate_audio(prompt, "speech_s2s", d=d, keys=S2S_KEYS[:-2], types=S2S_TYPES[:-2], topk=1)
    rr, rs = reference_loop(frames, TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
    assert torch.equal(r, rr) and torch.equal(s, rs)
    assert all(c[1] == 1 and c[2] == 0 for c in gen._model.calls) and gen.is_cfg


def _text_log(ids):
    """A scripted frame log whose text column spells `ids` then EOS (128001), audio columns zero (the on-device text loop)."""
    log = torch.zeros(len(ids) + 1 + 16, 1, 9, dtype=torch.int32)
    log[:len(ids), 0, 0] = torch.tensor(ids, dtype=torch.int32)
    log[len(ids):, 0, 0] = 128001
    return log


def _restated_asr_prompt(prompt, reason, sem):
    """prepare_lyric_asr_task / prepare_audio_caption_task restated (lyric_asr_task.py:175-200, audio_music_caption_task.py:175-200:
    identical bodies): task prompt rows (text column, mask on it) | reason bos, frames, eos | semantic bos, frames, eos, all
    shifted by the reason cardinality (audio columns, mask on them)."""
    P, w = prompt.shape[0], 9
    text = torch.zeros(P, w, dtype=torch.int64); text[:, -1] = prompt
    tmask = torch.zeros(P, w); tmask[:, -1] = 1
    r = torch.cat([torch.full((1, 8), TA.reason_bos), reason, torch.full((1, 8), TA.reason_eos)])
    sm = torch.cat([torch.full((1, 8), TA.semantic_bos), sem, torch.full((1, 8), TA.semantic_eos)]) + TA.audio_reason_card
    a = torch.cat([r, sm]).long()
    audio = torch.zeros(a.shape[0], w, dtype=torch.int64); audio[:, :8] = a
    amask = torch.zeros(a.shape[0], w); amask[:, :8] = 1
    return torch.cat([text, audio]), torch.cat([tmask, amask])


@pytest.mark.parametrize("module,prep,gen_name", [("lyric_asr_task", "prepare_lyric_asr_task", "generate_lyric_asr"),
                                                  ("audio_music_caption_task", "prepare_audio_caption_task", "generate_audio_caption"),
                                                  ("asr_task", "prepare_asr_task", "generate_asr")])
def test_asr_shaped_delegates_prompt_layout_and_loop(module, prep, gen_name):
    """The reference's lyric_asr_task / audio_music_caption_task Generators under their own names (the boundary: a direct
    importer of the reference's modules finds the same module, class and method names): prompt layout against the restated
    reference, text loop against a scripted model, generate_asr kept on all of them (the CLI calls it for lyric_recognition)."""
    import importlib
    Generator = importlib.import_module(f"uniaudio2_amd.evaluation.{module}").Generator
    ids = [11, 12, 13, 14, 15]
    gen = Generator(_ScriptedModel(_text_log(ids)), TA, text_tokenizer_path="ids")
    prompt = torch.tensor([128000, 5, 6, 128001])
    reason, sem = torch.arange(3 * 8).view(3, 8), 100 + torch.arange(4 * 8).view(4, 8)
    data, mask = getattr(gen, prep)(prompt, reason, sem)
    want_d, want_m = _restated_asr_prompt(prompt, reason, sem)
    assert torch.equal(data.long(), want_d) and torch.equal(mask.float(), want_m)
    out = getattr(gen, gen_name)(prompt, module, reason_token=reason, semantic_token=sem, topk=1)
    assert out == gen._text_tokenizer.decode(torch.tensor(ids))
    assert all(c[1] == 1 and c[2] == 1 for c in gen._model.calls)          # one sequence, text-feedback mode
    assert hasattr(gen, "generate_asr")


def test_audio_understanding_and_speech_s2t_delegates():
    """audio_understanding.py:233-339 / speech_s2t.py:274-381: the condition sequence (question text with its special tokens, then
    the clip's reason / semantic codes) and generate_answer; speech_s2t takes the codes in either orientation, returns
    (text, 1) and refuses prompts of >= 1500 frames with (-1, -1) as the reference does."""
    from uniaudio2_amd.evaluation import audio_understanding, speech_s2t
    ids = [21, 22, 23]
    reason, sem = torch.arange(8 * 3).view(8, 3), torch.arange(8 * 5).view(8, 5)
    prompt = torch.tensor([128000, 7, 128001])
    au = audio_understanding.Generator(_ScriptedModel(_text_log(ids)), TA, text_tokenizer_path="ids")
    q = torch.tensor([128000, 31, 32, 128001])
    d = {"text_seq_question": q, "reason_seq": reason, "semantic_seq": sem}
    data, mask = au.get_condition_seq(d, list(d), ["text", "audio", "audio"], prompt)
    nq = au.add_special_token("text_seq_question", q).shape[0]
    assert data.shape == (3 + nq + 5 + 7, 9)
    off = 3 + nq
    assert data[off, :8].eq(TA.reason_bos).all() and torch.equal(data[off + 1:off + 4, :8], reason.t())
    assert torch.equal(data[off + 6:off + 11, :8], sem.t() + TA.audio_reason_card)
    assert mask[:off, -1].eq(1).all() and mask[off:, :8].eq(1).all()
    assert au.generate_answer(prompt, "audio_understanding", d=d, keys=list(d), types=["text", "audio", "audio"], topk=1) == \
        au._text_tokenizer.decode(torch.tensor(ids))
    st = speech_s2t.Generator(_ScriptedModel(_text_log(ids)), TA, text_tokenizer_path="ids")
    d8 = {"reason_seq": reason, "semantic_seq": sem}
    dT = {"reason_seq": reason.t().contiguous(), "semantic_seq": sem.t().contiguous()}          # (T, 8), as the reference's CLI passes them
    a, am = st.get_condition_seq(d8, list(d8), ["audio", "audio"], prompt)
    b, bm = st.get_condition_seq(dT, list(dT), ["audio", "audio"], prompt)
    assert torch.equal(a, b) and torch.equal(am, bm) and a.shape == (3 + 5 + 7, 9)
    res = st.generate_answer(prompt, "speech_s2t", d=dT, keys=list(dT), types=["audio", "audio"], topk=5)
    assert res == (st._text_tokenizer.decode(torch.tensor(ids)), 1)
    # an 8-frame clip: the CLI hands over (T, 8) = (8, 8); speech_s2t.py:291-293 leaves a square input alone, i.e. takes it as (T, 8)
    r8 = torch.arange(64).view(8, 8)                                                             # (8 codebooks, T = 8) as stored on disk
    d88 = {"reason_seq": r8.t().contiguous(), "semantic_seq": sem.t().contiguous()}
    c, _ = st.get_condition_seq(d88, list(d88), ["audio", "audio"], prompt)
    assert torch.equal(c[3 + 1:3 + 1 + 8, :8], r8.t())                                           # frame t carries codebooks 0..7 of time t
    long_d = {"reason_seq": torch.zeros(8, 700, dtype=torch.long), "semantic_seq": torch.zeros(8, 800, dtype=torch.long)}
    assert st.generate_answer(prompt, "speech_s2t", d=long_d, keys=list(long_d), types=["audio", "audio"]) == (-1, -1)


def test_speech_edit_delegate_and_cli_routing():
    """speech_edit_ss.py:229-343 is speech_s2s's generator under another module name; the CLI routes every task to the module the
    reference routes it to (multi_task_inference.py:187-255)."""
    from uniaudio2_amd.evaluation import speech_edit_ss, speech_s2s
    from uniaudio2_amd.multi_task_inference import _get_generator_class
    assert issubclass(speech_edit_ss.Generator, speech_s2s.Generator) and hasattr(speech_edit_ss.Generator, "generate_audio")
    want = {"asr": "asr_task", "yue_asr": "asr_task", "lyric_recognition": "lyric_asr_task", "audio_caption": "audio_music_caption_task",
            "music_caption": "audio_music_caption_task", "audio_understanding": "audio_understanding", "speech_s2t": "speech_s2t",
            "tts": "tts_task", "yue_tts": "tts_task", "tta": "audiogen_task", "ttm": "musicgen_task", "lts": "songen_task",
            "instruct_tts": "insturct_tts_task", "speech_s2s": "speech_s2s"}
    for task, mod in want.items():
        assert _get_generator_class(task).__module__ == f"uniaudio2_amd.evaluation.{mod}", task
    with pytest.raises(ValueError):
        _get_generator_class("nope")


def test_ragged_schedule_runs_every_sequence_exactly_its_frames():
    """Continuous-batching plan of Model_stage3.generate_ragged."""
    from uniaudio2_amd.llm_models.model_new import ragged_schedule
    import random
    rnd = random.Random(0)
    for case in ([5], [3, 3, 3], [0, 2, 0], [60, 300, 61, 299, 60], [rnd.randint(0, 40) for _ in range(64)]):
        done = [0] * len(case)
        rows = list(range(len(case)))                    # rows[r] = original index of the sequence in row r
        for step, active, keep in ragged_schedule(case):
            assert active == rows and step >= 0
            for b in active:
                done[b] += step
            assert all(done[active[r]] < case[active[r]] for r in keep)                    # survivors still have work
            assert all(done[b] == case[b] for r, b in enumerate(active) if r not in keep)   # the retired are exactly done
            rows = [rows[r] for r in keep]
        assert done == list(case) and rows == []


class _ScriptedBatchModel:
    """CPU stand-in for Model_stage3's batched interface: every sequence replays its own scripted id stream; rows follow
    the retire_rows permutations exactly as the device state would."""

    def __init__(self, streams):
        self.streams = streams                      # per sequence: (F_b, 9) int32
        self._p = torch.nn.Parameter(torch.zeros(1))
        self._st = None
        self.rows, self.cursor, self.calls, self.retired = [], {}, [], []

    def parameters(self):
        return iter([self._p])

    def setup_caches(self, b, max_rows=64, log_frames=512):
        self._st = {"B": b, "max_rows": max_rows, "log_frames": log_frames}

    def begin_ragged(self, prompts):
        assert len(prompts) <= self._st["B"]
        self.prompt_lens = [int(t.shape[0]) for t, _ in prompts]
        self.rows = list(range(len(prompts)))
        self.cursor = {b: 0 for b in self.rows}

    def set_sampling(self, topk, temperature, seed=None):
        self.sampling = (topk, temperature)

    def generate_frames(self, n, batch, mode, reason_eos=-1, reason_card=0, max_pos=None, skip_text_head=False, skip_audio_experts=False):
        assert batch == len(self.rows) and mode == 0
        self.calls.append((n, batch))
        out = torch.zeros(n, batch, 9, dtype=torch.int32)
        for r, b in enumerate(self.rows):
            s = self.streams[b][self.cursor[b]:self.cursor[b] + n]
            out[:s.shape[0], r] = s
            self.cursor[b] += n
        return out

    def retire_rows(self, keep, batch):
        assert batch == len(self.rows)
        self.retired.append([self.rows[r] for r in range(batch) if r not in keep])
        self.rows = [self.rows[r] for r in keep]


def test_batched_tts_retires_sequences_at_their_own_eos():
    """generate_tts_batch: one ragged prefill, chunked decode of the live rows, per-sequence phase / EOS bookkeeping and
    retirement; every sequence's (reason, semantic) equals what the single-utterance loop returns for its stream."""
    from uniaudio2_amd.evaluation.tts_task import Generator
    g = torch.Generator().manual_seed(11)
    shapes = [(5, 9), (16, 16), (3, 40), (30, 2), (2, 2)]             # (reason frames, semantic frames) per utterance
    streams, frames_list = [], []
    for n_reason, n_sem in shapes:
        frames = [torch.randint(0, 4096, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_reason)]
        frames.append(torch.full((1, 8), TA.reason_eos, dtype=torch.int32))
        frames += [torch.randint(4100, 12292, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_sem)]
        frames.append(torch.full((1, 8), TA.semantic_eos + TA.audio_reason_card, dtype=torch.int32))
        frames += [torch.randint(0, 100, (1, 8), generator=g, dtype=torch.int32) for _ in range(48)]
        log = torch.zeros(len(frames), 9, dtype=torch.int32)
        log[:, 1:] = torch.cat(frames)
        streams.append(log); frames_list.append(frames)
    model = _ScriptedBatchModel(streams)
    gen = Generator.__new__(Generator)
    for k, v in vars(TA).items():
        setattr(gen, k, v)
    gen.empty_token, gen.is_cfg, gen._model, gen.device = 0, False, model, torch.device("cpu")
    gen.special_token_dict = gen.get_special_token()
    texts = [torch.arange(3 + i) for i in range(len(shapes))]         # different prompt lengths
    out = gen.generate_tts_batch(torch.tensor([128000, 1, 128001]), "tts", texts, topk=1)
    assert model._st["B"] >= len(shapes) and model.prompt_lens == [3 + 2 + 3 + i for i in range(len(shapes))]
    for (r, s), frames in zip(out, frames_list):
        rr, rs = reference_loop(frames, TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
        assert torch.equal(r, rr) and torch.equal(s, rs)
    # the batch shrank as utterances ended: 5 rows at first, fewer later, never a row computed after its retirement
    sizes = [b for _, b in model.calls]
    assert sizes[0] == 5 and sizes == sorted(sizes, reverse=True) and sizes[-1] < 5
    assert sorted(b for group in model.retired for b in group) == [0, 1, 2, 3, 4]
    assert model.sampling == (1, 0.9)


class _ScriptedPairModel(_ScriptedBatchModel):
    """As _ScriptedBatchModel, for the classifier-free-guidance pair layout: utterance b owns rows (2b, 2b + 1); the scripted
    stream is what the conditional row logs, the unconditional row logs the same ids (both continue from the pair's sample)."""

    def begin_ragged(self, prompts):
        assert len(prompts) % 2 == 0 and len(prompts) <= self._st["B"]
        self.prompt_lens = [int(t.shape[0]) for t, _ in prompts]
        self.prompts = prompts
        self.rows = list(range(len(prompts)))                   # row -> original row; utterance = row // 2
        self.cursor = {b: 0 for b in range(len(prompts) // 2)}

    def generate_frames(self, n, batch, mode, reason_eos=-1, reason_card=0, max_pos=None, skip_text_head=False, skip_audio_experts=False):
        assert batch == len(self.rows) and mode == 2 and batch % 2 == 0
        assert all(self.rows[r] + 1 == self.rows[r + 1] and self.rows[r] % 2 == 0 for r in range(0, batch, 2)), "pairs stay adjacent"
        self.calls.append((n, batch))
        out = torch.zeros(n, batch, 9, dtype=torch.int32)
        for r in range(0, batch, 2):
            u = self.rows[r] // 2
            s = self.streams[u][self.cursor[u]:self.cursor[u] + n]
            out[:s.shape[0], r] = s
            out[:s.shape[0], r + 1] = s
            self.cursor[u] += n
        return out


def test_batched_tts_with_classifier_free_guidance_keeps_row_pairs():
    """round 4: generate_tts_batch under --use_cfg — utterance b = rows (2b, 2b + 1) = (prompt, its all-pad twin of the same
    shape, tts_task.py:175-205), frames in mode 2, bookkeeping from the conditional row, a finished utterance retires BOTH rows
    and the survivors' pairs stay adjacent; results equal the single-utterance loop's."""
    from uniaudio2_amd.evaluation.tts_task import Generator
    g = torch.Generator().manual_seed(12)
    shapes = [(4, 7), (12, 3), (2, 20)]
    streams, frames_list = [], []
    for n_reason, n_sem in shapes:
        frames = [torch.randint(0, 4096, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_reason)]
        frames.append(torch.full((1, 8), TA.reason_eos, dtype=torch.int32))
        frames += [torch.randint(4100, 12292, (1, 8), generator=g, dtype=torch.int32) for _ in range(n_sem)]
        frames.append(torch.full((1, 8), TA.semantic_eos + TA.audio_reason_card, dtype=torch.int32))
        frames += [torch.randint(0, 100, (1, 8), generator=g, dtype=torch.int32) for _ in range(48)]
        log = torch.zeros(len(frames), 9, dtype=torch.int32)
        log[:, 1:] = torch.cat(frames)
        streams.append(log); frames_list.append(frames)
    model = _ScriptedPairModel(streams)
    gen = Generator.__new__(Generator)
    for k, v in vars(TA).items():
        setattr(gen, k, v)
    gen.empty_token, gen.is_cfg, gen._model, gen.device = 0, True, model, torch.device("cpu")
    gen.special_token_dict = gen.get_special_token()
    texts = [torch.arange(3 + i) for i in range(len(shapes))]
    out = gen.generate_tts_batch(torch.tensor([128000, 1, 128001]), "tts", texts, topk=1)
    # prompts went in as (conditional, unconditional) pairs of equal shape; the twin is all text padding
    assert model.prompt_lens == [n for i in range(len(shapes)) for n in (3 + 2 + 3 + i, 3 + 2 + 3 + i)]
    for p in range(len(shapes)):
        assert bool((model.prompts[2 * p + 1][0][:, -1] == TA.text_pad_token).all()) and not bool((model.prompts[2 * p][0][:, -1] == TA.text_pad_token).all())
    for (r, s), frames in zip(out, frames_list):
        rr, rs = reference_loop(frames, TA.reason_eos, TA.semantic_eos, TA.audio_reason_card)
        assert torch.equal(r, rr) and torch.equal(s, rs)
    sizes = [b for _, b in model.calls]
    assert sizes[0] == 6 and all(b % 2 == 0 for b in sizes) and sizes == sorted(sizes, reverse=True) and sizes[-1] < 6
    assert sorted(b for group in model.retired for b in group) == [0, 1, 2, 3, 4, 5]
    # a missing or mis-shaped twin is refused before anything runs
    with pytest.raises(ValueError):
        gen._generate_audio_tokens_batch([gen.prepare_tts_task(torch.tensor([128000, 1, 128001]), texts[0])], cfg_prompts=None)


def test_attention_row_groups_cover_every_row_once():
    """ops.attn_groups (host logic of the MFMA flash attention): every query row appears in exactly one group slot, groups hold
    rows of one sequence in position order, nkeys = 1 + the largest position of the group — for the default tile counts and for
    the 8-tile grouping the DiT's dense attention asks for."""
    import numpy as np
    import torch
    from uniaudio2_amd import ops
    rng = np.random.default_rng(3)
    lens = [150, 1, 77, 64, 129]
    seq = np.concatenate([np.full(n, b) for b, n in enumerate(lens)])
    pos = np.concatenate([rng.permutation(n) for n in lens])
    perm = rng.permutation(len(seq))
    seq, pos = seq[perm], pos[perm]
    for n_head, n_kv, q_tiles, want in ((24, 8, None, 2), (24, 24, None, 4), (24, 24, 8, 8)):
        rows, gseq, nkeys, qt = ops.attn_groups(pos, seq, n_head, n_kv, torch.device("cpu"), q_tiles=q_tiles)
        assert qt == want and rows.shape[1] == 16 * want
        rows, gseq, nkeys = rows.numpy(), gseq.numpy(), nkeys.numpy()
        live = rows[rows >= 0]
        assert sorted(live.tolist()) == list(range(len(seq)))
        for g in range(rows.shape[0]):
            r = rows[g][rows[g] >= 0]
            assert len(r) > 0 and np.all(seq[r] == gseq[g])
            assert np.all(np.diff(pos[r]) > 0) and nkeys[g] == pos[r].max() + 1
            assert np.all(rows[g][len(r):] == -1)            # padding only behind the live rows
