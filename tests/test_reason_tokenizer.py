"""Windowing arithmetic of token2audio_no_reason: pure host logic, restated reference loop."""
import math

import numpy as np
import pytest
import torch

from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import crossfade_concat, tile_codes, window_plan


def reference_indices(T, duration=20, rate=12.5, sr=24000):
    """reason_tokenizer.py:239-262,267 restated on a length only."""
    min_samples = int(duration * rate); hop = min_samples // 4 * 3; ov = min_samples - hop
    target = int(T / 12.5 * sr)
    x = list(range(T))
    if len(x) < min_samples:
        while len(x) < min_samples:
            x = x + x
        x = x[:min_samples]
    if (len(x) - ov) % hop > 0:
        n = math.ceil((len(x) - ov) / float(hop)) * hop + ov
        while len(x) < n:
            x = x + x
        x = x[:n]
    return x, list(range(0, len(x) - hop, hop)), target


@pytest.mark.parametrize("T", [1, 50, 249, 250, 251, 348, 436, 437, 500, 1000])
def test_window_plan_matches_reference_arithmetic(T):
    p = window_plan(T)
    x, starts, target = reference_indices(T)
    assert p["tiled_len"] == len(x) and p["starts"] == starts and p["target_len"] == target
    assert (p["min_codes"], p["hop_codes"], p["ovlp_codes"]) == (250, 186, 64)
    tiled = tile_codes(torch.arange(T).view(1, 1, T), p["tiled_len"])
    assert tiled.view(-1).tolist() == x
    assert (p["wav_window"], p["wav_hop"], p["wav_ovlp"]) == (480000, 360000, 120000)


def test_crossfade_is_linear_and_crops():
    a, b = torch.ones(1, 1000), torch.zeros(1, 1000)
    out = crossfade_concat([a, b], 1000, 250, 1600)
    assert out.shape == (1, 1600) and out[0, :750].eq(1).all() and out[0, 1000:].eq(0).all()
    np.testing.assert_allclose(out[0, 750:1000].numpy(), 1 - np.linspace(0, 1, 250), atol=1e-7)
