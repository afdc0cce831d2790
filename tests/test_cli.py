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
