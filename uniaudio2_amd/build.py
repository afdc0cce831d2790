"""Builds libua2hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels with
the source tree to the GPU box.  No torch, no cmake: `hipcc -shared -fPIC` on the .hip files.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libua2hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("UA2_EXTRA_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "ua2hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    hdr_t = max(os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "ua2hip.h")])
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        # per-object: a source is recompiled when it, or any header, is newer than its object (force = everything)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        procs.append((src, subprocess.Popen([HIPCC, *FLAGS, "-c", src, "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src}\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
