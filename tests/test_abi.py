"""The C-ABI library loads and exports every symbol include/ua2hip.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from uniaudio2_amd import _lib
    header = open(os.path.join(ROOT, "include", "ua2hip.h")).read()
    declared = set(re.findall(r"\b(ua2_[a-z0-9_]+)\s*\(", header))
    declared -= {"ua2_linear_args", "ua2_attn_args"}
    assert declared, "header parse failed"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in ua2hip.h but not exported by libua2hip.so"
    assert set(_lib.exported_symbols()) == declared, (set(_lib.exported_symbols()) ^ declared)
    assert _lib.lib.ua2_version() == 10


def test_packed_size_and_struct_layout():
    from uniaudio2_amd import _lib
    assert _lib.lib.ua2_packed_elems(_lib.UA2_BF16, 5120, 3072) == 5120 * 3072
    assert _lib.lib.ua2_packed_elems(_lib.UA2_BF16, 110, 128) == 112 * 128      # N padded to 16
    assert _lib.lib.ua2_packed_elems(_lib.UA2_F32, 16, 40) == 16 * 48           # K padded to 16
    # the ctypes mirrors have the C structs' sizes (also enforced at import)
    for i, st in enumerate(_lib.ABI_STRUCTS):
        assert _lib.lib.ua2_struct_size(i) == __import__("ctypes").sizeof(st), st.__name__
    assert _lib.lib.ua2_struct_size(99) == 0
    # the C side rejects a NULL args struct loudly instead of crashing
    assert _lib.lib.ua2_linear(None, None) != 0
    assert b"NULL" in _lib.lib.ua2_last_error()
