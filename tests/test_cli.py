"""CLI mirror: flag surface and the validation errors of the reference's main() (multi_task_inference.py:601-649), no GPU."""
import pytest

from uniaudio2_amd import multi_task_inference as cli

REFERENCE_FLAGS = ["task", "stage", "audio", "audio_dir", "reason_pt", "semantic_pt", "question", "question_file", "text", "text_file",
                   "output_dir", "results", "token_dir", "wav_dir", "prompt_text", "prompt_json", "llm_train_config", "resume",
                   "exp_dir", "text_tokenizer_path", "audio_tokenizer_config", "audio_model_path", "use_cfg", "temperature", "topk",
                   "cfg_scale", "decode_type", "codec_config", "codec_ckpt", "music_ssl_folder", "codec_steps", "codec_duration",
                   "seed", "rank"]


def test_every_reference_flag_exists_with_reference_defaults():
    a = cli.get_parser().parse_args(["--task", "TTS"])
    for f in REFERENCE_FLAGS:
        assert hasattr(a, f), f
    assert (a.temperature, a.topk, a.cfg_scale, a.seed, a.codec_steps, a.stage, a.decode_type) == (0.9, 50, 1.0, 888, 50, "all", "greedy")


@pytest.mark.parametrize("argv,msg", [
    (["--task", "TTS"], "provide --text or --text_file"),
    (["--task", "TTS", "--text", "hi"], "Set --llm_train_config and --text_tokenizer_path"),
    (["--task", "ASR"], "For understanding task provide"),
    (["--task", "bogus"], "Unsupported task"),
])
def test_validation_errors_match_reference(argv, msg):
    with pytest.raises(ValueError, match=msg):
        cli.main(argv)


def test_prompt_key_mapping():
    assert cli._prompt_key_from_task("yue_tts") == "Yue_TTS" and cli._prompt_key_from_task("tts") == "TTS"
    assert cli._prompt_key_from_task("speech_s2t") == "speech_s2t"


def test_token_files_follow_the_reference_contract(tmp_path):
    """multi_task_inference.py:522-523: `{name}_reason.pt` / `{name}_semantic.pt` = torch.save of (8, T) int32 CPU tensors
    (readers apply .long() / .transpose(0, 1), :304-308); the optional safetensors twin holds the same tensors."""
    import types
    import torch
    from safetensors.torch import load_file
    from uniaudio2_amd import multi_task_inference as cli
    args = types.SimpleNamespace(output_dir=str(tmp_path), save_safetensors=True)
    reason = torch.randint(0, 4096, (8, 7), dtype=torch.int32)
    semantic = torch.randint(0, 8192, (8, 19), dtype=torch.int32)
    cli._save_tokens(args, "utt_0", reason, semantic)
    r = torch.load(tmp_path / "utt_0_reason.pt", map_location="cpu")
    s = torch.load(tmp_path / "utt_0_semantic.pt", map_location="cpu")
    assert r.dtype == torch.int32 and r.shape == (8, 7) and torch.equal(r, reason) and torch.equal(s, semantic)
    assert r.transpose(0, 1).long().shape == (7, 8)
    st = load_file(str(tmp_path / "utt_0_tokens.safetensors"))
    assert torch.equal(st["reason"], reason) and torch.equal(st["semantic"], semantic)
    args.save_safetensors = False
    cli._save_tokens(args, "utt_1", reason, semantic)
    assert not (tmp_path / "utt_1_tokens.safetensors").exists() and (tmp_path / "utt_1_reason.pt").exists()


def test_extension_flags_default_to_the_reference_behaviour():
    """The MI355X-only flags give up nothing unless asked: --codec_batch 1 = the reference's one-by-one stage-2 loop
    (multi_task_inference.py:540-548: a waveform never depends on its batch-mates), --order_free_rows 0 = every LM row keeps the
    bits of its single-sequence run, --batch_size 1 = one utterance per GPU at a time."""
    a = cli.get_parser().parse_args(["--task", "TTS"])
    assert (a.codec_batch, a.order_free_rows, a.batch_size, a.dtype) == (1, 0, 1, "bf16")
    b = cli.get_parser().parse_args(["--task", "TTS", "--codec_batch", "8", "--order_free_rows", "256"])
    assert (b.codec_batch, b.order_free_rows) == (8, 256)


def test_stage2_shard_by_length_is_a_balanced_partition():
    """Stage 2 deals utterances longest first (as stage 1 does): every rank's list is sorted by length, so consecutive --codec_batch
    groups share a window count, and the ranks' total frames differ by at most one utterance."""
    names = [f"n{i:02d}" for i in range(23)]
    frames = [250 * (1 + (7 * i) % 3) + i for i in range(23)]               # 1-, 2- and 3-window utterances, interleaved by name
    for world in (1, 2, 4, 8):
        shards = [cli.stage2_shard(names, world, r, lengths=frames) for r in range(world)]
        assert sorted(sum(shards, [])) == names
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
        for sh in shards:
            ln = [frames[names.index(n)] for n in sh]
            assert ln == sorted(ln, reverse=True)
        tot = [sum(frames[names.index(n)] for n in sh) for sh in shards]
        assert max(tot) - min(tot) <= max(frames)
