#!/usr/bin/env python3
"""Fixture for BASELINE.json config 1: the samples of the reference's samples/p225_002.wav (22 050 Hz IEEE-float32 mono,
86 848 samples, 3.94 s) as a data file, so the codec plumbing test can run where /root/reference does not exist.
Data only — no reference source.  Container-only.  Usage: python tests/golden/make_golden_p225.py"""
import os

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
sr, x = wavfile.read("/root/reference/samples/p225_002.wav")
x = np.asarray(x)
assert x.ndim == 1 and x.dtype == np.float32, (x.shape, x.dtype)
np.savez_compressed(os.path.join(HERE, "p225_002.npz"), samples=x, sample_rate=np.int32(sr))
print(sr, x.shape, x.dtype, float(np.abs(x).max()))
