"""The persistent depth-decoder engine prototype (tools/ubench/engine.hip; profiles/r5_notes.md §8) restates ua2_linear's B = 1
arithmetic from scratch — LDS-DMA loader, MFMA consumers with two half-chunks per instruction, per-range partial sums through LDS,
granule hand-offs between CUs — and must reproduce the 17 production launches of a depth-decoder pass BIT FOR BIT: an
independent second implementation of csrc/ua2_gemv.hip's summation-order contract (lit_model.py:424,511,591-595 at one row)."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_prototype_is_bit_identical_to_the_launch_chain():
    lib = os.path.join(ROOT, "tools", "ubench", "libengine.so")
    if not os.path.exists(lib):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("libengine.so not built and no hipcc here")
        subprocess.run(["bash", os.path.join(ROOT, "tools", "ubench", "build_engine.sh")], check=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ubench", "engine_run.py"), "--iters", "5", "--timeout-ms", "200"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "give-up code 0x0" in out, out[-3000:]
    assert "engine == chain, every op, bit for bit" in out, out[-3000:]
    assert out.count("bit-identical") == 17 and "DIFFERENT" not in out, out[-3000:]
