#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage for one .hip file, as a table (registers, scratch, occupancy per kernel).
Usage: python tools/resource_usage.py uniaudio2_amd/csrc/ua2_skinny.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/ru.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.splitlines():
    if "error" in l:
        print(l)
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"n": m.group(1)}
        rows.append(cur)
    for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "LDS Size", "SGPRs"):
        m = re.search(k + r"[^:]*: (\d+)", l)
        if m and cur is not None and k not in cur:
            cur[k] = m.group(1)
names = subprocess.run(["c++filt"], input="\n".join(r["n"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n.replace("void (anonymous namespace)::", "").replace("void ", ""))
    if flt in n:
        print(f"{n:60s} V={r.get('VGPRs')} A={r.get('AGPRs')} scratch={r.get('ScratchSize')} occ={r.get('Occupancy')} S={r.get('SGPRs')}")
