"""Pins oracle/rvq_oracle.c against codes produced by the reference's vendored core_vq (golden)."""
import os

import numpy as np
import pytest

from make_golden_rvq import CASES, make_inputs
from oracle import rvq_oracle


@pytest.mark.parametrize("name", list(CASES))
def test_rvq_oracle_matches_reference_codes(golden_dir, name):
    d = np.load(os.path.join(golden_dir, "rvq_toy.npz"))
    c = CASES[name]
    x, emb = make_inputs(c)                                # (B, D, T), (L, C, D)
    flat = x.permute(0, 2, 1).reshape(-1, c["D"]).numpy()  # rows = (b, t)
    codes, q, margin = rvq_oracle.rvq_encode(flat, emb.numpy(), want_margin=True)
    ref = np.ascontiguousarray(d[f"{name}_codes"].reshape(c["L"], -1).T)         # (N, L)
    # identical wherever the distance gap to the runner-up is above fp32 noise; the reference's
    # cdist-based distances cannot be reproduced bit for bit (different expansion), so allow a
    # mismatch only on a level whose margin is < 1e-4 relative, and nothing after it for that row
    bad = 0
    for n in range(codes.shape[0]):
        for l in range(c["L"]):
            if codes[n, l] != ref[n, l]:
                assert margin[n, l] < 1e-4 * max(1.0, float(np.abs(flat[n]).max())), (n, l, margin[n, l])
                bad += 1
                break
    assert bad <= 1, f"{bad} rows diverged from the reference"
    if c["dup"]:
        assert not np.isin(codes, [7, 100]).any(), "duplicate codewords must resolve to the lower index"
        assert not np.isin(ref, [7, 100]).any()
    same = (codes == ref).all(1)
    dec = rvq_oracle.rvq_decode(ref.astype(np.int32), emb.numpy())
    ref_dec = np.transpose(d[f"{name}_decoded"], (0, 2, 1)).reshape(-1, c["D"])
    np.testing.assert_allclose(dec, ref_dec, rtol=0, atol=1e-6)
    np.testing.assert_array_equal(q[same], dec[same])       # encode's quantised sum == lookup of its own codes
