#!/usr/bin/env python3
"""Golden vectors for the tokenizer's host logic, produced by RUNNING THE REFERENCE's own methods
`ReasoningTokenizer.token2audio_no_reason` (tools/tokenizer/ReasoningCodec_film/reason_tokenizer.py:229-306) and
`ReasoningTokenizer.audio2token` (:86-129) — unbound, on a SimpleNamespace `self` — with the neural pieces around them
(model.inference_codes, SQCodec.decode, model.fetch_codes_batch, the Whisper front end) replaced by the deterministic
stand-ins of tokenizer_stub.py.  What the reference's code does here and the fixture records: the 250-code windows with
hop 186 (code indices of every window), the in-context latent chain (how many frames, which values), the torch.randn
draw order and shapes (seeded CPU generator), the float64 cross-fade and the final crop; for audio2token the
self-concatenation / segment grid / chunking by batch_size / token crop and the order of the time_film draws.

Absent third-party imports of reason_tokenizer.py:1-20 (omegaconf, torchaudio, transformers, the codec model modules) are
MagicMock'ed: none of them is executed by the two methods.  Container-only (needs /root/reference).
Usage: python tests/golden/make_golden_tokenizer.py"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from tokenizer_stub import (CLIP_CASES, SEED, T_CASES, WAVE_STRIDE, StubCodec, StubEncoderModel, StubModel, make_clip, make_codes,
                            wave_digest)

P = "tools.tokenizer.ReasoningCodec_film."
STUBBED = ["omegaconf", "torchaudio", "torchaudio.transforms", "transformers", "huggingface_hub", "safetensors", "safetensors.torch",
           P + "models.AudioDiffusion1D", P + "models.model_utils", P + "models.scalar24k", "tools.tokenizer.common"]


def main():
    for name in STUBBED:
        sys.modules[name] = MagicMock()
    rt = importlib.import_module(P + "reason_tokenizer")
    RT = rt.ReasoningTokenizer
    out = {}
    torch.set_num_threads(4)

    # ---- token2audio_no_reason -------------------------------------------------------------------------------------------
    for T in T_CASES:
        model, codec = StubModel(), StubCodec()
        fake = types.SimpleNamespace(device=torch.device("cpu"), sample_rate=24000, rec_frame_rate=12.5, reason_frame_rate=5,
                                     sq_codec_hz=25, model=model, SQCodec=codec)
        torch.manual_seed(SEED)
        wave = RT.token2audio_no_reason(fake, make_codes(T), False, duration=20, guidance_scale=1.5, num_steps=7, disable_progress=True)
        k = f"t2a_{T}_"
        out[k + "windows"] = torch.stack([c["codes"][0] for c in model.calls]).numpy().astype(np.int32)      # (n_win, 8, 250)
        out[k + "incontext"] = np.array([c["incontext"] for c in model.calls], dtype=np.int32)
        out[k + "latent_length"] = np.array([c["latent_length"] for c in model.calls], dtype=np.int32)
        # the random part of every call's `true_latents` (frames >= incontext) pins the draw order; keep a thin slice of it
        out[k + "noise"] = np.stack([c["true"][0, c["incontext"]:c["incontext"] + 3, :5].numpy() for c in model.calls])
        out[k + "noise_tail"] = np.stack([c["true"][0, -2:, -5:].numpy() for c in model.calls])
        out[k + "ctx_head"] = np.stack([c["true"][0, :2, :5].numpy() for c in model.calls])                 # in-context frames (calls >= 1)
        out[k + "wave_shape"] = np.array(wave.shape, dtype=np.int64)
        out[k + "wave_sub"] = wave[0, ::WAVE_STRIDE].numpy().astype(np.float32)
        out[k + "wave_digest"] = wave_digest(wave)
        assert wave.dtype == torch.float32
        print("token2audio_no_reason T =", T, "windows", len(model.calls), "wave", tuple(wave.shape), "incontext", out[k + "incontext"].tolist())

    # ---- audio2token -----------------------------------------------------------------------------------------------------
    for n, bs in CLIP_CASES:
        model = StubEncoderModel()
        fake = types.SimpleNamespace(device=torch.device("cpu"), sample_rate=24000, rec_frame_rate=12.5, reason_frame_rate=5, model=model)
        fake.get_whisper_features = lambda audio, sr: torch.zeros(audio.shape[0], 80, 8)
        torch.manual_seed(SEED)
        reason, rec = RT.audio2token(fake, make_clip(n, 900 + n % 97), 24000, False, batch_size=bs)
        k = f"a2t_{n}_"
        out[k + "reason"] = reason.numpy().astype(np.int32)
        out[k + "rec"] = rec.numpy().astype(np.int32)
        out[k + "chunk_rows"] = np.array([c["rows"] for c in model.fetch_calls], dtype=np.int32)
        out[k + "masks"] = np.concatenate([torch.stack(c["masks"]).numpy() for c in model.fetch_calls], axis=1).astype(np.uint8)   # (3, total rows)
        print("audio2token n =", n, "reason", tuple(reason.shape), "rec", tuple(rec.shape), "chunks", out[k + "chunk_rows"].tolist())
    np.savez_compressed(os.path.join(HERE, "tokenizer_host.npz"), **out)
    print("wrote", os.path.join(HERE, "tokenizer_host.npz"), os.path.getsize(os.path.join(HERE, "tokenizer_host.npz")), "bytes")


if __name__ == "__main__":
    main()
