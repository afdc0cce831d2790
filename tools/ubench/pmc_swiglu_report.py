#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/ubench/pmc_swiglu.py -> the per-launch HBM traffic table of the roofline kernel.
Usage: pmc_swiglu_report.py fetch.db write.db"""
import sqlite3, sys


def rows(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for did, name, cn, v, dur in db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection order by dispatch_id"):
        if "gemv_kernel" in name and cn == counter:
            k = out.setdefault(did, [0.0, dur])
            k[0] += v
    return [out[k] for k in sorted(out)]


f, w = rows(sys.argv[1], "FETCH_SIZE"), rows(sys.argv[2], "WRITE_SIZE")
ALG = 2 * 3072 * 8192 * 2 + 3072 * 2 + 3072 // 16 * 4 + 8192 * 4
print("# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/ubench/pmc_swiglu.py   (MI355X; a second pass with --pmc WRITE_SIZE)")
print("# kernel gemv_kernel<1,4,2,4,true>: scaled-RMSNorm + fc_1/fc_2 + SwiGLU GEMV, 3072 -> 2x8192, bf16, M=1 (the form the B = 1 frame runs); 8 launches, 8 distinct weight sets")
print("# FETCH_SIZE unit = KiB; on gfx950 it reports 1/2 of a wide coalesced stream (MI355X_MICROARCH.md §HBM) -> corrected = 2 x value")
tot = 0.0
for v, dur in f:
    b = 2 * v * 1024
    tot += b
    print(f"FETCH_SIZE_KiB {v:.1f}  duration_us(profiled) {dur / 1e3:.2f}  corrected_bytes {int(b)}")
mean = tot / max(1, len(f))
print(f"# mean corrected HBM read traffic per launch: {mean / 1e6:.2f} MB; algorithmic bytes per launch: {ALG / 1e6:.2f} MB; ratio {mean / ALG:.4f}")
print(f"# WRITE_SIZE (KiB, uncalibrated width): {[round(v, 1) for v, _ in w]}")
print(f"# average profiled duration {sum(d for _, d in f) / max(1, len(f)) / 1e3:.2f} us")
print(f"PMC_SWIGLU_TRAFFIC {int(mean)}")
