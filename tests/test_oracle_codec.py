"""Pins oracle/codec_oracle.py (CPU restatement of the live codec's conv auto-encoder and window arithmetic) on outputs
of the reference itself (tests/golden/codec_toy.npz, produced by importing and running the reference's ScalarModel),
and checks the product's pure window-plan function against the oracle's line-by-line restatement."""
import os

import numpy as np
import pytest
import torch

from make_golden_codec import SCALAR_CFG, codec_state_dict
from weights import seeded_tensor


def _toy_sd():
    import json
    here = os.path.join(os.path.dirname(__file__), "golden")
    meta = json.load(open(os.path.join(here, "codec_toy.json")))
    shapes = {k: tuple(s) for k, s in meta["scalar_keys"]}
    return codec_state_dict(shapes, 21)


def test_scalar_oracle_matches_reference_outputs(golden_dir):
    from oracle.codec_oracle import ScalarOracle
    d = np.load(os.path.join(golden_dir, "codec_toy.npz"))
    o = ScalarOracle(_toy_sd(), SCALAR_CFG)
    wav = seeded_tensor((2, 1, 16 * 130 + 5), 31, std=0.3)
    lat = o.encode(wav).numpy()
    assert lat.shape == d["scalar_latent"].shape
    assert np.abs(lat - d["scalar_latent"]).max() < 2e-6          # same torch ops on the same CPU: thread-count noise only
    rec = o.decode(torch.from_numpy(d["scalar_latent"])).numpy()
    assert rec.shape == d["scalar_wav"].shape
    assert np.abs(rec - d["scalar_wav"]).max() < 2e-6 * max(1.0, float(np.abs(d["scalar_wav"]).max()))


@pytest.mark.parametrize("T", [1, 7, 63, 64, 187, 249, 250, 251, 348, 436, 437, 500, 622, 623, 1000])
def test_window_plan_of_the_product_equals_line_by_line_restatement(T):
    """The product's closed-form plan (tiled length + window starts) must read exactly the code indices the reference's
    loop reads (oracle: index lists built by literally concatenating and slicing as reason_tokenizer.py:251-267 does)."""
    from oracle.codec_oracle import window_indices
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import tile_codes, window_plan
    want = window_indices(T)
    plan = window_plan(T)
    codes = torch.arange(T).view(1, 1, T)
    tiled = tile_codes(codes, plan["tiled_len"])
    got = [tiled[0, 0, s:s + plan["min_codes"]].tolist() for s in plan["starts"]]
    assert got == want["windows"]
    assert (plan["ovlp_frames"], plan["target_len"], plan["wav_window"], plan["wav_ovlp"]) == \
           (want["ovlp_frames"], want["target_len"], want["wav_window"], want["wav_ovlp"])


def test_crossfade_matches_restatement():
    from oracle.codec_oracle import crossfade
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import crossfade_concat
    g = torch.Generator().manual_seed(5)
    segs = [torch.randn(1, 480, generator=g) for _ in range(3)]
    a = crossfade_concat([s.clone() for s in segs], 400, 100, 950)
    b = crossfade([s.clone() for s in segs], 400, 100, 950)
    assert a.dtype == b.dtype and torch.equal(a, b) and a.shape == (1, 950)
