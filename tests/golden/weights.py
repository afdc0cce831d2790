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
