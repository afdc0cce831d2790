#!/usr/bin/env python3
"""Per-launch HBM-side traffic of ScalarModel.decode from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes:
they do not fit one) of `python tools/ubench/codec_decode.py`:  python tools/ubench/pmc_codec.py fetch.db write.db
Prints the conv launches of the LAST decode of each pass side by side, with the gfx950 correction of MI355X_MICROARCH.md §HBM
(FETCH_SIZE tallies a 128-byte request of a wide coalesced read as 64 B -> x2) and the algorithmic bytes of bench.py."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def launches(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection order by dispatch_id").fetchall()
    by = {}
    for did, name, cn, v, dur in rows:
        if cn == counter and ("conv" in name or "tc_pack" in name):
            k = by.setdefault(did, [name, 0.0, dur])
            k[1] += v
    seq = [by[k] for k in sorted(by)]
    n = 45 if len(seq) % 45 == 0 else 44
    per = [seq[i:i + n] for i in range(0, len(seq), n)]
    return per[-1]


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:58]


def main():
    import bench
    f, w = launches(sys.argv[1], "FETCH_SIZE"), launches(sys.argv[2], "WRITE_SIZE")
    assert len(f) == len(w), (len(f), len(w))
    print(f"{'kernel':60s} {'FETCH_KiB':>10s} {'x2 (gfx950)':>12s} {'WRITE_KiB':>10s} {'us':>8s}")
    tf = tw = tus = 0.0
    for (n1, fv, d1), (n2, wv, d2) in zip(f, w):
        print(f"{short(n1):60s} {fv:10.0f} {2 * fv:12.0f} {wv:10.0f} {d1 / 1e3:8.1f}")
        tf += fv; tw += wv; tus += d1 / 1e3
    alg = bench.scalar_decode_work(bench.SCALAR_CFG, 500)["bytes"]
    print(f"\nsum over the {len(f)} launches of one decode: FETCH {tf / 1024:.1f} MiB raw = {2 * tf * 1024 / 1e9:.3f} GB corrected, WRITE {tw * 1024 / 1e9:.3f} GB, "
          f"kernel time {tus:.0f} us")
    print(f"algorithmic bytes (bench.py scalar_decode_work): {alg / 1e9:.3f} GB  ->  traffic / algorithmic = {(2 * tf + tw) * 1024 / alg:.2f} "
          f"(raw, uncorrected reads: {(tf + tw) * 1024 / alg:.2f})")


if __name__ == "__main__":
    main()
