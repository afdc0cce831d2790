"""ctypes binding of libua2hip.so — the stub a maintainer of the reference would add
(INTEGRATION.md).  Mirrors include/ua2hip.h one to one; no torch types cross the boundary.

There is no CPU fallback: if the shared library is missing or fails to load, importing
this module raises (the product must fail loudly, never silently run something else).
"""
import ctypes as C
import os

# torch owns the device memory and the streams this library is handed, so both must sit on ONE HIP
# runtime: importing torch first makes the dynamic loader resolve libua2hip.so's libamdhip64 to the copy
# torch already loaded (loading ours first gives the process two runtimes and launches fail with
# "no ROCm-capable device").
import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UA2_LIB", os.path.join(_HERE, "libua2hip.so"))   # UA2_LIB: timing experiments load an instrumented build (tools/ubench)

UA2_F32, UA2_BF16 = 0, 1
PRO_CAST, PRO_NORM, PRO_LOCAL_ATTN = 0, 1, 3
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU, EPI_QKV_ROPE, EPI_GELU = 0, 1, 2, 3, 4
NORM_RMS_LIT, NORM_RMS_MOSHI, NORM_LAYERNORM = 0, 1, 2
ROPE_HALF_SPLIT, ROPE_INTERLEAVED, ROPE_NONE = 0, 1, 2
ACT_KIND_DEFAULT, GELU_TANH, GATE_SIGMOID_SECOND = 0, 1, 2
SUM_ORDER_INVARIANT, SUM_ORDER_FREE = 0, 1
ATTN_BF16_QP = 1
UA2_PAGE = 64

vp, i32, f32, i64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64


class KvGeom(C.Structure):
    _fields_ = [("k_pool", vp), ("v_pool", vp), ("page_table", vp), ("max_pages", i32), ("n_kv", i32),
                ("n_head", i32), ("head_size", i32), ("ring_pages", i32)]


class LinearArgs(C.Structure):
    _fields_ = [("dtype", i32), ("prologue", i32), ("epilogue", i32), ("M", i32), ("N", i32), ("K", i32),
                ("x", vp), ("ldx", i32), ("norm_w", vp), ("eps", f32),
                ("w0", vp), ("w1", vp), ("y", vp), ("ldy", i32), ("resid", vp), ("ldr", i32),
                ("part_max", vp), ("part_idx", vp), ("forbid", vp), ("row_pos", vp), ("row_seq", vp),
                ("rope_cos", vp), ("rope_sin", vp), ("q_out", vp), ("kv", KvGeom),
                ("norm_b", vp), ("norm_kind", i32), ("out_scale", vp), ("rope_mode", i32),
                ("workspace", vp), ("workspace_bytes", C.c_size_t), ("y_packed", vp), ("x_packed", vp),
                ("bias", vp), ("bias1", vp), ("act_kind", i32),
                ("y_norm_w", vp), ("y_h", vp), ("ldh", i32), ("y_ssq", vp), ("x_h", vp), ("x_ssq", vp),
                ("split_ws", vp), ("split_ws_bytes", C.c_size_t), ("sum_order", i32),
                ("y_ln_w", vp), ("y_ln_b", vp), ("y_ln_eps", f32),
                ("range_ws", vp), ("range_ws_bytes", C.c_size_t)]


class AttnArgs(C.Structure):
    _fields_ = [("dtype", i32), ("R", i32), ("q", vp), ("row_pos", vp), ("row_seq", vp), ("kv", KvGeom), ("y", vp),
                ("window", i32), ("y_packed", vp), ("group_rows", vp), ("group_seq", vp), ("group_nkeys", vp), ("n_groups", i32),
                ("group_q_tiles", i32), ("flags", i32)]


class Conv1dArgs(C.Structure):
    _fields_ = [("B", i32), ("Cin", i32), ("Cout", i32), ("Tin", i32), ("Tout", i32), ("K", i32), ("stride", i32),
                ("dilation", i32), ("pad_left", i32), ("in_repeat", i32), ("out_phases", i32), ("out_trim_left", i32),
                ("pre_act", i32), ("post_act", i32), ("x", vp), ("w", vp), ("bias", vp), ("pre_alpha", vp),
                ("post_alpha", vp), ("post_alpha_n", i32), ("residual", vp), ("y", vp), ("w_lo", vp), ("precision", i32), ("w2", vp), ("w2_lo", vp), ("bias2", vp), ("alpha2", vp)]


class ConvTcArgs(C.Structure):
    _fields_ = [("B", i32), ("Cin", i32), ("Cout", i32), ("Tin", i32), ("Tout", i32), ("K", i32), ("dilation", i32),
                ("pad_left", i32), ("in_repeat", i32), ("out_phases", i32), ("out_trim_left", i32), ("post_act", i32),
                ("variant", i32), ("x_hi", vp), ("x_lo", vp), ("w", vp), ("w_lo", vp), ("bias", vp), ("post_alpha", vp),
                ("post_alpha_n", i32), ("w2", vp), ("w2_lo", vp), ("bias2", vp), ("alpha2", vp), ("res_hi", vp), ("res_lo", vp),
                ("y_hi", vp), ("y_lo", vp), ("y_f32", vp)]


ACT_NONE, ACT_PRELU, ACT_ELU, ACT_TANH, ACT_ROUND9 = 0, 1, 2, 3, 4
EW_IDENTITY, EW_SILU, EW_SIGMOID, EW_TANH = 0, 1, 2, 3


class GptDesc(C.Structure):
    _fields_ = [("n_layer", i32), ("n_embd", i32), ("n_head", i32), ("n_kv", i32), ("head_size", i32),
                ("inter", i32), ("eps", f32),
                ("qkv", C.POINTER(vp)), ("proj", C.POINTER(vp)), ("fc1", C.POINTER(vp)), ("fc2", C.POINTER(vp)),
                ("mlp_proj", C.POINTER(vp)), ("norm1", C.POINTER(vp)), ("norm2", C.POINTER(vp)),
                ("ln_f", vp), ("rope_cos", vp), ("rope_sin", vp),
                ("k_pool", C.POINTER(vp)), ("v_pool", C.POINTER(vp)), ("page_table", vp), ("max_pages", i32)]


class Stage3Desc(C.Structure):
    _fields_ = [("dtype", i32), ("n_cb", i32), ("va", i32), ("vt", i32), ("max_rows", i32), ("max_batch", i32),
                ("und", GptDesc), ("backbone", GptDesc), ("gen", GptDesc), ("decoder", GptDesc),
                ("wte", vp), ("audio_emb", vp), ("lm_head", vp), ("projection", vp), ("audio_head", C.POINTER(vp)),
                ("tokens", vp), ("mask", vp), ("row_pos", vp), ("row_seq", vp), ("dec_pos", vp), ("dec_seq", vp),
                ("forbid", vp), ("out_tokens", vp), ("frame_log", vp), ("counters", vp), ("log_frames", i32),
                ("scratch", vp), ("scratch_floats", C.c_size_t)]


_EXPORTS = {
    "ua2_last_error": (C.c_char_p, []),
    "ua2_version": (C.c_int, []),
    "ua2_packed_elems": (C.c_size_t, [C.c_int, i64, i64]),
    "ua2_pack_linear": (C.c_int, [vp, C.c_int, C.c_int, i64, i64, vp, C.c_int, C.c_int, vp]),
    "ua2_linear": (C.c_int, [C.POINTER(LinearArgs), vp]),
    "ua2_linear_order_free_accepts": (C.c_int, [C.POINTER(LinearArgs)]),
    "ua2_debug_force_general_linear": (C.c_int, [C.c_int]),
    "ua2_debug_kernel_launches": (i64, [C.c_char_p]),
    "ua2_debug_refresh_env": (None, []),
    "ua2_struct_size": (C.c_size_t, [C.c_int]),
    "ua2_dwconv1d": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ua2_cfg_mix": (C.c_int, [vp, i32, i32, C.c_float, vp, vp, vp, i32, vp]),
    "ua2_stage3_set_cfg": (C.c_int, [vp, C.c_float]),
    "ua2_stage3_set_order_free_rows": (C.c_int, [vp, i32]),
    "ua2_linear_workspace_bytes": (C.c_size_t, [C.c_int, i64, i64]),
    "ua2_linear_chain_timed": (C.c_int, [C.POINTER(LinearArgs), i32, i32, vp, C.POINTER(C.c_float)]),
    "ua2_attn": (C.c_int, [C.POINTER(AttnArgs), vp]),
    "ua2_attn_local": (C.c_int, [C.POINTER(AttnArgs), vp]),
    "ua2_embed_frame": (C.c_int, [C.c_int, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ua2_rmsnorm_blend": (C.c_int, [i32, i32, vp, vp, f32, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "ua2_argmax_embed": (C.c_int, [C.c_int, i32, i32, vp, vp, vp, i32, i32, vp, i32, i32, vp, vp]),
    "ua2_conv1d": (C.c_int, [C.POINTER(Conv1dArgs), vp]),
    "ua2_conv1d_tc": (C.c_int, [C.POINTER(ConvTcArgs), vp]),
    "ua2_tc_pack": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "ua2_tc_unpack": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "ua2_avgpool1d": (C.c_int, [vp, vp, i64, i32, i32, vp]),
    "ua2_rvq_encode": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, C.c_size_t, vp]),
    "ua2_rvq_workspace_bytes": (C.c_size_t, [i64, i32]),
    "ua2_rvq_fallbacks": (C.c_int, [vp]),
    "ua2_rvq_decode": (C.c_int, [vp, vp, i64, i32, i32, i32, vp, vp]),
    "ua2_ew_fma": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, f32, f32, vp]),
    "ua2_ew_act": (C.c_int, [vp, vp, i64, i32, vp]),
    "ua2_gather_rows": (C.c_int, [vp, vp, vp, i64, i32, vp]),
    "ua2_time_film": (C.c_int, [vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "ua2_layernorm_rows": (C.c_int, [vp, vp, vp, vp, i64, i32, f32, vp]),
    "ua2_qknorm_rope_kv": (C.c_int, [C.c_int, vp, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, C.POINTER(KvGeom), vp]),
    "ua2_stage3_scratch_floats": (C.c_size_t, [C.POINTER(Stage3Desc)]),
    "ua2_stage3_create": (C.c_int, [C.POINTER(Stage3Desc), C.POINTER(vp)]),
    "ua2_stage3_destroy": (None, [vp]),
    "ua2_sample_topk": (C.c_int, [C.c_int, i32, vp, i32, i32, i32, f32, vp, C.c_uint64, vp, i32, vp, i32, i32, vp, i32, i32, vp, i32, vp]),
    "ua2_stage3_set_sampling": (C.c_int, [vp, i32, f32, C.c_uint64, vp]),
    "ua2_stage3_set_prefill_groups": (C.c_int, [vp, vp, vp, vp, i32, i32]),
    "ua2_stage3_trunk": (C.c_int, [vp, i32, vp]),
    "ua2_stage3_heads": (C.c_int, [vp, i32, vp]),
    "ua2_stage3_feedback": (C.c_int, [vp, i32, i32, i32, i32, vp]),
    "ua2_stage3_frame": (C.c_int, [vp, i32, i32, i32, i32, i32, vp]),
    "ua2_stage3_buffer": (C.POINTER(C.c_float), [vp, C.c_char_p]),
}


ABI_STRUCTS = (KvGeom, LinearArgs, AttnArgs, Conv1dArgs, GptDesc, Stage3Desc, ConvTcArgs)


def exported_symbols():
    """Every symbol include/ua2hip.h declares (checked by the CPU test-suite)."""
    return sorted(_EXPORTS)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m uniaudio2_amd.build or "
            "__graft_entry__.build()).  uniaudio2_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _EXPORTS.items():
        fn = getattr(lib, name)           # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    for i, st in enumerate(ABI_STRUCTS):    # a stale .so or a drifted binding must not corrupt kernel arguments silently
        if lib.ua2_struct_size(i) != C.sizeof(st):
            raise ImportError(f"{LIB_PATH}: sizeof({st.__name__}) is {lib.ua2_struct_size(i)} in the library but "
                              f"{C.sizeof(st)} in the ctypes binding; rebuild (python -m uniaudio2_amd.build)")
    return lib


lib = _load()


class Ua2Error(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise Ua2Error(f"{what} failed ({rc}): {lib.ua2_last_error().decode(errors='replace')}")
