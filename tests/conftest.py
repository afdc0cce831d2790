import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no libua2hip.so (built artefacts are git-ignored): build it once (hipcc cross-compiles
    # gfx950 without a GPU).  Only when the file is MISSING — the GPU box receives the library with the tree and must not
    # spend its minutes recompiling because of copied timestamps.
    lib = os.path.join(ROOT, "uniaudio2_amd", "libua2hip.so")
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location("ua2_build", os.path.join(ROOT, "uniaudio2_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_lib(force=True, verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fresh_kernel_env(request):
    """The launchers cache their UA2_* knobs (ua2hip.h ua2_debug_refresh_env).  Tests change the environment through monkeypatch /
    os.environ; this fixture is set up first and torn down last, so the cache is re-read before a GPU test starts and after every
    other fixture (monkeypatch included) has restored the environment."""
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu:
        from uniaudio2_amd._lib import lib
        lib.ua2_debug_refresh_env()
    yield
    if gpu:
        from uniaudio2_amd._lib import lib
        lib.ua2_debug_refresh_env()
