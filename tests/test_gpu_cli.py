"""`--stage 2` / the second half of `--stage all` end to end through the CLI on synthetic checkpoints (SURVEY.md §8f #1:
"--stage all completes on synthetic weights"): the codec is loaded from files laid out as the reference's are (infer yaml ->
sqcodec yaml + .pth, DiT json, model checkpoint with 'module.' prefixes and foreign SSL keys), every `*_semantic.pt` of the
token dir is decoded to a wav file."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from codec_model_stub import module_state_dict
from make_golden_codec import codec_state_dict
from weights import seeded_tensor

pytestmark = pytest.mark.gpu


def test_cli_stage2_decodes_token_files_with_a_synthetic_codec(tmp_path):
    from scipy.io import wavfile
    from test_gpu_codec import BENCH_SCALAR_CFG
    from uniaudio2_amd import multi_task_inference as cli
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.AudioDiffusion1D import AudioDiffusion1D
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    D, L = 256, BENCH_SCALAR_CFG["latent_hidden_dim"]
    dit = dict(num_attention_heads=4, attention_head_dim=64, in_channels=2 * L + D, out_channels=L, num_layers=2, norm_type="ada_norm_single",
               activation_fn="gelu-approximate", attention_bias=True, norm_eps=1e-6, cross_attention_dim=None, _class_name="Transformer1DModel")
    json.dump(dit, open(tmp_path / "model_config.json", "w"))
    sq = ScalarModel(**BENCH_SCALAR_CFG)
    torch.save({"codec_model": codec_state_dict({k: tuple(v.shape) for k, v in sq.state_dict().items()}, 77)}, tmp_path / "sqcodec.pth")
    yaml.safe_dump({"generator": {"config": BENCH_SCALAR_CFG}}, open(tmp_path / "sqcodec_config.yaml", "w"))
    m = AudioDiffusion1D(unet_model_config_path=dit, whisper_fea_dim=64, wavlm_fea_dim=96, codec_dim=D, encoder_depth=1)
    sd = module_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 909)
    for k in list(sd):
        if k.endswith("_codebook.embed"):
            sd[k] = seeded_tensor(sd[k].shape, len(k), std=0.5)
    ckpt = {"module." + k: v for k, v in sd.items()}
    ckpt["module.whisper_encoder.layers.0.fc1.weight"] = torch.zeros(4, 4)           # frozen SSL weights ride in the released checkpoint: ignored
    torch.save({"model": ckpt}, tmp_path / "codec.checkpoint")
    yaml.safe_dump(dict(sq_config=str(tmp_path / "sqcodec_config.yaml"), sq_resume=str(tmp_path / "sqcodec.pth"),
                        transformer_diffusion_config=str(tmp_path / "model_config.json"), num_channels=32, whisper_path="unused"),
                   open(tmp_path / "infer_config.yaml", "w"))
    tok = tmp_path / "tokens"
    tok.mkdir()
    g = torch.Generator().manual_seed(3)
    for name, T in (("utt_0", 30), ("utt_1", 300)):                                   # one window; two cross-faded windows
        torch.save(torch.randint(0, 4096, (8, T // 2), generator=g, dtype=torch.int32), tok / f"{name}_reason.pt")
        torch.save(torch.randint(0, 8192, (8, T), generator=g, dtype=torch.int32), tok / f"{name}_semantic.pt")
    torch.save(torch.zeros(8, 3, dtype=torch.int32), tok / "orphan_reason.pt")        # no semantic twin: skipped (:541-543)
    cli.main(["--task", "TTS", "--stage", "2", "--text", "unused", "--llm_train_config", "unused", "--text_tokenizer_path", "unused",
              "--prompt_text", "unused", "--token_dir", str(tok), "--codec_config", str(tmp_path / "infer_config.yaml"),
              "--codec_ckpt", str(tmp_path / "codec.checkpoint"), "--codec_steps", "2", "--seed", "1"])
    for name, T in (("utt_0", 30), ("utt_1", 300)):
        sr, x = wavfile.read(tok / "wavs" / f"{name}.wav")
        assert sr == 24000 and x.dtype == np.int16 and x.shape == (int(T / 12.5 * 24000),)
        assert np.abs(x.astype(np.int32)).max() > 0
    assert not (tok / "wavs" / "orphan.wav").exists()
