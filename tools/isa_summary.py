#!/usr/bin/env python3
"""Memory / MFMA / wait skeleton of one kernel's ISA: python tools/isa_summary.py file.hip 'mangled-name-substring' [max lines]"""
import re, subprocess, sys
src, key = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", "/tmp/isa.s"],
               capture_output=True)
s = open("/tmp/isa.s").read()
m = re.search(r"^(_Z\S*" + re.escape(key) + r"\S*):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M)
body = m.group(2).splitlines()
print(m.group(1), len(body), "lines")
pat = r"global_load|buffer_load|s_waitcnt|v_mfma|s_barrier|ds_write|ds_read|s_cbranch|global_store|scratch_|^\.LBB"
out, prev, cnt = [], None, 0
for l in body:
    if not re.search(pat, l.strip()):
        continue
    k = re.sub(r"\s+", " ", l.strip())
    k = re.sub(r"v\[\d+:\d+\]|v\d+|s\[\d+:\d+\]|s\d+|a\[\d+:\d+\]", "R", k)
    k = re.sub(r"offset:\d+", "off", k)
    k = re.sub(r";.*", "", k)
    if k == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{cnt:3d}x {prev}")
        prev, cnt = k, 1
out.append(f"{cnt:3d}x {prev}")
print("\n".join(out[:n]))
