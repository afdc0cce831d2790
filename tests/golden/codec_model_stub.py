"""Seeded inputs, weights and the stand-in flow estimator shared by tests/golden/make_golden_codec_model.py (which runs
the reference) and the tests (which never touch it).  Test data plumbing — not product code, not reference code."""
from collections import OrderedDict
from types import SimpleNamespace

import torch

from weights import seeded_tensor

CFG = dict(D=256, Cw=64, Cl=96, Cb=1024, depth=2, latent=24, steps=4, B=3, T25=30, film_seed=11,
           infer_B=1, infer_T=10, infer_card=32, latent_length=16, incontext=6)


def module_state_dict(shapes, seed):
    """Name- and shape-driven synthetic checkpoint: fan-in scaled matrices, weight-norm gains / norm weights around 1,
    LayerScale large enough to matter (the reference initialises it at 1e-2), small biases."""
    out = OrderedDict()
    for i, (k, shp) in enumerate(shapes.items()):
        t = seeded_tensor(shp, seed * 7919 + i, std=1.0)
        if k.endswith("inv_freq"):
            n = shp[0]
            t = 1.0 / (10000 ** (torch.arange(0, 2 * n, 2).float() / (2 * n)))       # RotaryEmbedding buffer (modules/transformer.py:107)
        elif k.endswith("original0") or k.endswith("weight_g"):
            t = 1.0 + 0.2 * t
        elif k.endswith(".scale"):
            t = 0.5 + 0.1 * t
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * t
        elif k.endswith("bias"):
            t = 0.1 * t
        elif k.endswith("cls_token") or k.endswith("scale_shift_table") or "zero_cond" in k:
            t = 0.5 * t
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = t / max(fan_in, 1) ** 0.5
        out[k] = t
    return out


def think_inputs():
    c = CFG
    return seeded_tensor((c["B"], c["Cw"], 2 * c["T25"]), 401, std=1.0), seeded_tensor((c["B"], c["Cb"], c["T25"]), 402, std=1.0)


def fetch_inputs():
    c = CFG
    return dict(whisper=seeded_tensor((c["B"], c["Cw"], 2 * c["T25"]), 411, std=1.0),
                wavlm=seeded_tensor((c["B"], c["Cl"], 2 * c["T25"]), 412, std=1.0),
                bestrq_acoustic=seeded_tensor((c["B"], c["Cb"], c["T25"]), 413, std=1.0),
                bestrq_semantic=seeded_tensor((c["B"], c["Cb"], c["T25"]), 414, std=1.0))


def infer_inputs():
    c = CFG
    B, T, D, L = c["infer_B"], c["infer_T"], c["D"], c["latent"]
    g = torch.Generator().manual_seed(421)
    return dict(codes=torch.randint(0, c["infer_card"], (B, 8, T), generator=g),
                tab_phone=seeded_tensor((1, c["infer_card"], D), 422, std=0.5), tab_sem=seeded_tensor((1, c["infer_card"], D), 423, std=0.5),
                tab_ac=seeded_tensor((6, c["infer_card"], D), 424, std=0.3), noise=seeded_tensor((B, 2 * T, L), 425, std=1.0),
                first_latent=seeded_tensor((B, 2 * T, L), 426, std=1.0), true_latent=seeded_tensor((B, 2 * T, L), 427, std=1.0),
                zero_cond=seeded_tensor((D,), 428, std=0.5), latent_length=c["latent_length"], incontext=c["incontext"])


class StubEstimator:
    """Deterministic stand-in for the DiT inside BASECFM (AudioDiffusion1D.py:106-121 calls
    `estimator(x, timestep=..., added_cond_kwargs=...).sample`): mixes channels, time steps and the timestep."""

    def __init__(self, device="cpu"):
        c = CFG
        self.W = seeded_tensor((2 * c["latent"] + c["D"], c["latent"]), 431, std=(2 * c["latent"] + c["D"]) ** -0.5).to(device)

    def __call__(self, x, timestep=None, added_cond_kwargs=None):
        L = CFG["latent"]
        y = torch.tanh(x.float() @ self.W) + 0.3 * timestep.float().view(-1, 1, 1) * x[..., :L].float()
        y = y + 0.1 * torch.roll(x[..., L:2 * L].float(), 1, dims=1)
        return SimpleNamespace(sample=y)
