"""Pins oracle/codec_model_oracle.py on outputs of the reference's OWN code (tests/golden/codec_model_toy.npz, produced by
tests/golden/make_golden_codec_model.py, which imports models/AudioDiffusion1D.py + modules/transformer.py and runs
encode_reasoning_part, fetch_codes_batch, inference_codes and BASECFM.solve_euler).  What stays unpinned (diffusers DiT,
vector_quantize_pytorch) is listed in the oracle's header."""
import json
import os

import numpy as np
import pytest
import torch

from codec_model_stub import CFG, StubEstimator, fetch_inputs, infer_inputs, module_state_dict, think_inputs


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "codec_model_toy.npz")), json.load(open(os.path.join(golden_dir, "codec_model_toy.json")))


def _sd(meta, key, seed):
    return module_state_dict({k: tuple(s) for k, s in meta[key]}, seed)


def full_state_dict(meta):
    """AudioDiffusion1D-style state dict of the toy model: audio_thinking.* (seed 301) + the top-level layers (302)."""
    sd = {"audio_thinking." + k: v for k, v in _sd(meta, "think_keys", 301).items()}
    sd.update(_sd(meta, "fetch_keys", 302))
    return sd


def test_thinking_encoder_matches_reference(gold):
    from oracle.codec_model_oracle import encode_reasoning_query
    d, meta = gold
    w, m = think_inputs()
    q = encode_reasoning_query(_sd(meta, "think_keys", 301), w, m).numpy()
    assert q.shape == d["think_query"].shape
    assert np.abs(q - d["think_query"]).max() < 2e-5 * max(1.0, float(np.abs(d["think_query"]).max()))


def test_fetch_codes_pipeline_matches_reference(gold):
    from oracle.codec_model_oracle import fetch_codes_from_features
    d, meta = gold
    f = fetch_inputs()
    masks = torch.from_numpy(d["fetch_film_masks"])
    assert masks.any() and not masks.all(), "the golden must exercise both branches of the FiLM mask"
    r = fetch_codes_from_features(full_state_dict(meta), f["whisper"], f["wavlm"], f["bestrq_acoustic"], f["bestrq_semantic"], masks)
    for name, key in (("reason_query", "fetch_reason_query"), ("pre_vq_phone", "fetch_pre_vq_phone"), ("pre_vq_semantic", "fetch_pre_vq_semantic"),
                      ("pre_vq_acoustic", "fetch_pre_vq_acoustic"), ("merge_features", "fetch_merge_features")):
        got, ref = r[name].numpy(), d[key]
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() < 3e-5 * max(1.0, float(np.abs(ref).max())), name
    assert r["merge_codes"].shape[-1] == 3            # identity quantisers: one zero code column per group, order phone|semantic|acoustic


@pytest.mark.parametrize("tag", ["infer_first", "infer_other"])
def test_inference_codes_and_euler_match_reference(gold, tag):
    from oracle.codec_model_oracle import inference_codes
    d, meta = gold
    i = infer_inputs()
    cfe = _sd(meta, "cfe_keys", 303)
    est = StubEstimator()
    lookups = [lambda idx, t=t: sum(t[l][idx[..., l]] for l in range(idx.shape[-1])) for t in (i["tab_phone"], i["tab_sem"], i["tab_ac"])]
    true_lat, n_inc = (i["first_latent"], 0) if tag == "infer_first" else (i["true_latent"], i["incontext"])
    lat = inference_codes(lookups, cfe["weight"], cfe["bias"], i["zero_cond"], lambda x, t: est(x, timestep=t).sample, i["codes"],
                          true_lat.clone(), i["latent_length"], n_inc, i["noise"], guidance_scale=1.5, num_steps=CFG["steps"]).numpy()
    assert lat.shape == d[tag].shape
    assert np.abs(lat - d[tag]).max() < 1e-5 * max(1.0, float(np.abs(d[tag]).max()))


def test_dit_oracle_shapes_and_cfg_consistency():
    """The DiT restatement is parity-unpinned (diffusers absent); what can be checked without the package: shapes, batch
    independence (a row's output does not depend on the other batch element), and that zero-initialised gates reduce the
    network to proj_out(norm_out(proj_in(x) + pos))."""
    from oracle.codec_model_oracle import dit_forward
    import dit_toy
    sd = dit_toy.state_dict(5)
    x = torch.randn(2, 12, dit_toy.CFG["in_channels"], generator=torch.Generator().manual_seed(1))
    t = torch.tensor([0.3, 0.3])
    y = dit_forward(sd, x, t, dit_toy.CFG["heads"], dit_toy.CFG["head_dim"])
    assert y.shape == (2, 12, dit_toy.CFG["out_channels"]) and torch.isfinite(y).all()
    y0 = dit_forward(sd, x[:1], t[:1], dit_toy.CFG["heads"], dit_toy.CFG["head_dim"])
    assert torch.allclose(y[:1], y0, atol=1e-5)
