"""ctypes wrapper of oracle/rvq_oracle.c (CPU ORACLE — test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "librvq_oracle.so")


def _load():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "rvq_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    lib = C.CDLL(_LIB)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.rvq_encode_oracle.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_int, C.c_int, ip, fp, fp]
    lib.rvq_decode_oracle.argtypes = [ip, fp, C.c_int64, C.c_int, C.c_int, C.c_int, fp]
    return lib


def rvq_encode(x, emb, want_margin=False):
    """x [N,D] fp32, emb [L,C,D] fp32 -> codes [N,L] int32, quantized [N,D] (, margin [N,L])."""
    lib = _load()
    x = np.ascontiguousarray(x, np.float32); emb = np.ascontiguousarray(emb, np.float32)
    N, D = x.shape; L, Cc, D2 = emb.shape
    assert D == D2 and D <= 1024
    codes = np.empty((N, L), np.int32); q = np.empty((N, D), np.float32); mg = np.empty((N, L), np.float32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.rvq_encode_oracle(x.ctypes.data_as(fp), emb.ctypes.data_as(fp), N, L, Cc, D, codes.ctypes.data_as(ip),
                          q.ctypes.data_as(fp), mg.ctypes.data_as(fp))
    return (codes, q, mg) if want_margin else (codes, q)


def rvq_decode(codes, emb):
    lib = _load()
    codes = np.ascontiguousarray(codes, np.int32); emb = np.ascontiguousarray(emb, np.float32)
    N, L = codes.shape; L2, Cc, D = emb.shape
    assert L == L2
    out = np.empty((N, D), np.float32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.rvq_decode_oracle(codes.ctypes.data_as(ip), emb.ctypes.data_as(fp), N, L, Cc, D, out.ctypes.data_as(fp))
    return out
