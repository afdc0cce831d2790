"""Seeded, shape-driven synthetic weights shared by the golden generators and the tests.

The reference ships no checkpoints, so every parity case runs on
random weights.  The golden generators (run only in the authoring container, with
/root/reference importable) and the tests (run anywhere, never touching the reference)
must produce bit-identical weights from a seed: both call ``seeded_state_dict`` with the
same ordered ``{name: shape}`` mapping, which the golden file records.

This file is test data plumbing, not product code and not reference code.
"""
from collections import OrderedDict

import torch


def seeded_tensor(shape, seed, std=0.05, kind="normal"):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if kind == "norm":          # RMSNorm / PReLU-like gains: around 1 so they matter
        return 1.0 + 0.1 * t
    return std * t


def seeded_state_dict(shapes, seed, std=0.05):
    """shapes: ordered mapping name -> shape.  Returns OrderedDict name -> fp32 tensor.

    Each tensor gets its own generator seeded by (seed, index) so a change in one shape
    does not shift the stream of the others.
    """
    out = OrderedDict()
    for i, (name, shape) in enumerate(shapes.items()):
        is_norm = (".norm_" in name or "ln_f" in name) and name.endswith("weight")
        out[name] = seeded_tensor(shape, seed * 100003 + i, std=std,
                                  kind="norm" if is_norm else "normal")
    return out


def checksum(sd):
    """Order-independent fingerprint (float64 sums) to detect generator drift."""
    s = 0.0
    a = 0.0
    for k in sorted(sd):
        t = sd[k].double()
        s += float(t.sum())
        a += float(t.abs().sum())
    return [s, a]


def mimi_state_dict(shapes, seed):
    """Shape- and name-driven synthetic checkpoint for the MimiCodec family (make_golden_mimi.py and its tests):
    fan-in scaled weights, gains around 1, LayerScale large enough to matter, positive codebook usage counts."""
    out = OrderedDict()
    for i, (k, shp) in enumerate(shapes.items()):
        t = seeded_tensor(shp, seed * 7919 + i, std=1.0)
        if k.endswith("_initialized"):
            t = torch.ones(shp)
        elif k.endswith("cluster_usage"):
            t = 1.0 + 0.5 * t.abs()
        elif k.endswith("embedding_sum"):
            t = t
        elif k.endswith("scale"):
            t = 0.5 + 0.1 * t
        elif "norm" in k and (k.endswith("weight") or k.endswith("alpha")):
            t = 1.0 + 0.1 * t
        elif k.endswith("bias"):
            t = 0.1 * t
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = t / max(fan_in, 1) ** 0.5
        out[k] = t
    return out
