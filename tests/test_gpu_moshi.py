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
