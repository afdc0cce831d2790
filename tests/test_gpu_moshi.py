"""GPU parity of the Moshi-flavour transformer against outputs of the reference's importable twin (golden):
whole-sequence forward, fp32 kernels within 2e-4; bf16 kernels within 3e-2; incremental decoding == whole sequence."""
import json
import os

import numpy as np
import pytest
import torch

from make_golden_moshi import DEP, MIMI, SIN, moshi_state_dict
from weights import seeded_tensor

pytestmark = pytest.mark.gpu


def _build(cfg, seed):
    from uniaudio2_amd.tools.tokenizer.MimiCodec.model.modules.transformer import StreamingTransformer
    m = StreamingTransformer(**cfg)
    m.load_state_dict(moshi_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    return m.cuda()


@pytest.mark.parametrize("name,cfg,T,seed", [("mimi", MIMI, 50, 51), ("dep", DEP, 4, 52), ("sin", SIN, 20, 53)])
def test_streaming_transformer_vs_reference(golden_dir, name, cfg, T, seed):
    d = np.load(os.path.join(golden_dir, "moshi_toy.npz"))
    meta = json.load(open(os.path.join(golden_dir, "moshi_toy.json")))
    m = _build(cfg, seed)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == {k: s for k, s in meta[name + "_keys"]}
    x = seeded_tensor((2, T, cfg["d_model"]), seed + 100, std=1.0).cuda()
    m.prepare(max_batch=2, max_seq_length=64, dtype=torch.float32)
    y = m(x).cpu().numpy()
    np.testing.assert_allclose(y, d[name + "_out"], atol=2e-4, rtol=0)
    # incremental decoding (prefix, then one position at a time) reproduces the whole-sequence result
    m.prepare(max_batch=2, max_seq_length=64, dtype=torch.float32)
    half = T // 2
    parts = [m(x[:, :half].contiguous(), offset=0)] + [m(x[:, t:t + 1].contiguous(), offset=t) for t in range(half, T)]
    np.testing.assert_allclose(torch.cat(parts, 1).cpu().numpy(), d[name + "_out"], atol=2e-4, rtol=0)
    m.prepare(max_batch=2, max_seq_length=64, dtype=torch.bfloat16)
    yb = m(x).cpu().numpy()
    assert np.abs(yb - d[name + "_out"]).max() < 3e-2 * max(1.0, np.abs(d[name + "_out"]).max())


def test_streaming_session_equals_whole_sequence(golden_dir):
    """`with m.streaming(batch):` — chunks of 1..7 positions, the transformer keeping its own offset and cache
    (transformer.py:676-695) — reproduces the whole-sequence golden of the Mimi flavour (rope, context window) and of
    the sin_rope flavour."""
    d = np.load(os.path.join(golden_dir, "moshi_toy.npz"))
    for name, cfg, T, seed in (("mimi", MIMI, 50, 51), ("sin", SIN, 20, 53)):
        m = _build(cfg, seed)
        x = seeded_tensor((2, T, cfg["d_model"]), seed + 100, std=1.0).cuda()
        parts, t, step = [], 0, 1
        with m.streaming(2):
            while t < T:
                parts.append(m(x[:, t:t + step].contiguous()))
                t += step
                step = step % 7 + 1
        assert not m.is_streaming
        np.testing.assert_allclose(torch.cat(parts, 1).cpu().numpy(), d[name + "_out"], atol=2e-4, rtol=0)


def test_ring_cache_streams_past_the_linear_capacity():
    """RingKVCache (llm_modules/transformer.py:211-278) role: with a finite `context` a streaming session re-uses a fixed
    ring of cache pages, so it may run past any linear capacity.  2500 positions (> the 2048-slot linear cache round 1 capped
    sessions at) streamed in chunks of 1..150 equal the whole-sequence forward over a 2500-slot linear cache with the same
    context window — bit for bit: the ring only changes WHERE a key is stored, the attention visits the same keys in the
    same order.  (The reference ring's off-by-one, SURVEY A.16, is consciously not reproduced: streaming == whole sequence.)"""
    from uniaudio2_amd.tools.tokenizer.MimiCodec.model.modules.transformer import StreamingTransformer
    cfg = dict(MIMI, context=100)
    T = 2500
    whole_m = StreamingTransformer(**cfg)
    sd = moshi_state_dict({k: tuple(v.shape) for k, v in whole_m.state_dict().items()}, 51)
    whole_m.load_state_dict(sd)
    whole_m = whole_m.cuda()
    x = seeded_tensor((1, T, cfg["d_model"]), 777, std=1.0).cuda()
    whole_m.prepare(max_batch=1, max_seq_length=T, dtype=torch.float32)
    whole = whole_m(x)
    m = StreamingTransformer(**cfg)
    m.load_state_dict(sd)
    m = m.cuda()
    parts, t, step = [], 0, 1
    with m.streaming(1):
        while t < T:
            parts.append(m(x[:, t:t + step].contiguous()))
            t += step
            step = step * 3 % 151 + 1
        assert m._plan["ring_pages"] > 0 and m._plan["k"][0].shape[0] == m._plan["ring_pages"]      # a handful of pages, not T / 64
    got = torch.cat(parts, 1)
    assert got.shape == whole.shape and torch.equal(got, whole)
    # the ring plan of the finished session must not serve a later whole-sequence call (T = 2500 rows in one launch
    # would overwrite ring slots that earlier rows still attend to): forward() re-plans a linear cache
    again = m(x)
    assert m._plan["ring_pages"] == 0 and torch.equal(again, whole)
    with pytest.raises(RuntimeError):
        with m.streaming(1):
            m(x[:, :10].contiguous())
        m(x[:, 10:20].contiguous(), offset=10)      # explicit offset against a cache that is now a ring: refused, not silently wrong
