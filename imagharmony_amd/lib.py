"""ctypes binding of libimh_hip.so (include/imh.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load,
importing any op raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(hipcc --offload-arch=gfx950, in-tree).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMH_LIB_PATH overrides; IMH_EXPERIMENTAL=1 (the switch build.py compiles the experimental library under) selects that library by itself,
# so that the one variable builds AND loads the same file
_EXP_PATH = os.path.join(_HERE, "..", "tools", "tmp_libs", "libimh_hip_experimental.so")
LIB_PATH = os.environ.get("IMH_LIB_PATH") or (_EXP_PATH if os.environ.get("IMH_EXPERIMENTAL") == "1" else os.path.join(_HERE, "libimh_hip.so"))

ABI_VERSION = 9
IMH_DT_BF16, IMH_DT_F16 = 0, 1
GF_GEGLU, GF_ACT_GELU, GF_ACT_SILU, GF_VT_PERM, GF_OUT_F32, GF_LN_ROW, GF_LN_COL = 1, 2, 4, 8, 16, 32, 64
OP_GEMM, OP_ATTN, OP_GROUPNORM, OP_LAYERNORM, OP_EW, OP_ATTN_SMALL, OP_GEMM_DUAL, OP_XATTN = 0, 1, 2, 3, 4, 5, 6, 7
GN_ALL, GN_STATS, GN_TABLE, GN_APPLY, GN_TABLE_APPLY = 0, 1, 2, 3, 4
(EW_TIMESTEP, EW_SILU, EW_CONCAT, EW_CONV_IN, EW_CFG_STEP, EW_CAST_F32, EW_ADD, EW_STEP_SET, EW_CFG_RESCALE, EW_SOFTMAX,
 EW_ROW_STATS, EW_STEP_ROW) = range(12)

# variant codes / modes that only a -DIMH_EXPERIMENTAL build compiles (csrc/imh_common.h IMH_EXP_ONLY): measured, selected by no
# tuning.json entry and no default mode.  experimental() asks the loaded library (imh_debug_set(1, 0)).
EXP_VARIANTS = {(4064, 64), (4064, 128), (4128, 64), (5256, 320), (6128, 320), (6064, 160), (7064, 160), (256, 128), (256, 256), (22128, 160),
                (3128, 128), (24128, 128), (9256, 320), (7564, 320), (7564, 160), (7328, 160), (7428, 160), (7256, 80)}


def experimental():
    return load().imh_debug_set(1, 0) == 1


def variant_built(cfg):
    """cfg = (bm, bn[, splits]): is this tile variant in the loaded library?"""
    return (int(cfg[0]), int(cfg[1])) not in EXP_VARIANTS or experimental()


_i32, _f32, _vp = C.c_int32, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("X", _vp), ("W", _vp), ("Y", _vp), ("partial", _vp), ("bias", _vp), ("rowadd", _vp),
                ("residual", _vp), ("ln_s", _vp), ("ln_c", _vp), ("ln_eps", _f32),
                ("ln_stats", _vp), ("ln_stats_out", _vp), ("ln_slots", _i32), ("ln_slots_out", _i32),
                ("gn_out", _vp), ("gn_nblk", _i32), ("gn_hw", _i32), ("gn_tab", _vp), ("gn_silu", _i32),
                ("gn_part", _vp), ("gn_part2", _vp), ("gn_gamma", _vp), ("gn_beta", _vp), ("gn_eps", _f32), ("gn_groups", _i32), ("gn_pC1", _i32),
                ("gn_pnblk", _i32), ("gn_psub", _i32), ("gn_pnpart", _i32), ("gn_pnblk2", _i32), ("gn_psub2", _i32), ("gn_pnpart2", _i32),
                ("X2", _vp), ("Cin1", _i32), ("Yt", _vp), ("yt_col0", _i32), ("ldyt", _i32),
                ("M", _i32), ("N", _i32), ("K", _i32),
                ("ldx", _i32), ("ldw", _i32), ("ldy", _i32), ("ldr", _i32), ("ldra", _i32),
                ("rows_per_batch", _i32), ("splits", _i32), ("flags", _i32),
                ("H", _i32), ("Wd", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32), ("stride", _i32), ("up", _i32),
                ("dtype", _i32), ("conv", _i32), ("bm", _i32), ("bn", _i32), ("pf_ptr", _vp), ("pf_bytes", C.c_uint32), ("xcd", _i32)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", _vp), ("K", _vp), ("Vt", _vp), ("K2", _vp), ("Vt2", _vp), ("O", _vp),
                ("B", _i32), ("H", _i32), ("Lq", _i32), ("Lk", _i32), ("Lk_pad", _i32), ("Lk2", _i32),
                ("Lk2_pad", _i32),
                ("ldq", _i32), ("ldk", _i32), ("ldvt", _i32), ("ldk2", _i32), ("ldvt2", _i32), ("ldo", _i32),
                ("scale", _f32), ("scale2", _f32), ("scale2_tab", _vp), ("step", _vp), ("dtype", _i32),
                ("pf_ptr", _vp), ("pf_bytes", C.c_uint32)]


class XAttnArgs(C.Structure):
    _fields_ = [("X", _vp), ("Wq", _vp), ("ln_s", _vp), ("ln_c", _vp), ("ln_eps", _f32), ("ln_stats", _vp), ("ln_slots", _i32),
                ("K", _vp), ("Vt", _vp), ("K2", _vp), ("Vt2", _vp), ("O", _vp),
                ("B", _i32), ("H", _i32), ("Lq", _i32), ("C", _i32), ("Lk", _i32), ("Lk_pad", _i32), ("Lk2", _i32),
                ("Lk2_pad", _i32),
                ("ldx", _i32), ("ldw", _i32), ("ldk", _i32), ("ldvt", _i32), ("ldk2", _i32), ("ldvt2", _i32), ("ldo", _i32),
                ("scale", _f32), ("scale2", _f32), ("scale2_tab", _vp), ("step", _vp), ("dtype", _i32),
                ("pf_ptr", _vp), ("pf_bytes", C.c_uint32)]


class SmallAttnArgs(C.Structure):
    _fields_ = [("Q", _vp), ("K", _vp), ("V", _vp), ("O", _vp),
                ("B", _i32), ("H", _i32), ("Lq", _i32), ("Lk", _i32), ("dq", _i32), ("dv", _i32),
                ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32), ("scale", _f32), ("dtype", _i32)]


class NormArgs(C.Structure):
    _fields_ = [("x", _vp), ("y", _vp), ("gamma", _vp), ("beta", _vp), ("partial", _vp),
                ("B", _i32), ("HW", _i32), ("C", _i32), ("groups", _i32), ("rows", _i32),
                ("eps", _f32), ("silu", _i32), ("dtype", _i32),
                ("mode", _i32), ("table", _vp), ("partial2", _vp), ("nblk", _i32), ("sub", _i32), ("npart", _i32), ("C1", _i32),
                ("nblk2", _i32), ("sub2", _i32), ("npart2", _i32), ("pf_ptr", _vp), ("pf_bytes", C.c_uint32)]


class EwArgs(C.Structure):
    _fields_ = [("a", _vp), ("b", _vp), ("y", _vp), ("w", _vp), ("bias", _vp), ("tab", _vp), ("step", _vp),
                ("n", C.c_int64),
                ("i0", _i32), ("i1", _i32), ("i2", _i32), ("i3", _i32), ("i4", _i32), ("i5", _i32),
                ("f0", _f32), ("f1", _f32), ("f2", _f32), ("f3", _f32), ("dtype", _i32)]


class F32Args(C.Structure):
    _fields_ = [("X", _vp), ("W", _vp), ("Y", _vp), ("bias", _vp), ("residual", _vp), ("gamma", _vp), ("beta", _vp), ("ws", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32), ("ldx", _i32), ("ldw", _i32), ("ldy", _i32), ("ldr", _i32),
                ("conv", _i32), ("H", _i32), ("Wd", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32), ("up", _i32),
                ("B", _i32), ("HW", _i32), ("C", _i32), ("groups", _i32), ("nblk", _i32), ("silu", _i32),
                ("eps", _f32), ("scale", _f32)]


F32_GEMM, F32_GN_STATS, F32_GN_TABLE, F32_GN_APPLY, F32_SOFTMAX = range(5)

# every symbol include/imh.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("imh_abi_version", C.c_int, []),
    ("imh_last_error", C.c_char_p, []),
    ("imh_debug_set", C.c_int, [C.c_int, C.c_int]),
    ("imh_gemm", C.c_int, [C.POINTER(GemmArgs), _vp]),
    ("imh_gemm_dual", C.c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _vp]),
    ("imh_gemm_pick_config", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]),
    ("imh_gemm_workspace_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("imh_gemm_stats_slot_width", C.c_int, [C.c_int, C.c_int]),
    ("imh_gemm_gn_block_rows", C.c_int, [C.c_int, C.c_int]),
    ("imh_attention", C.c_int, [C.POINTER(AttnArgs), _vp]),
    ("imh_cross_attention", C.c_int, [C.POINTER(XAttnArgs), _vp]),
    ("imh_attention_small", C.c_int, [C.POINTER(SmallAttnArgs), _vp]),
    ("imh_groupnorm", C.c_int, [C.POINTER(NormArgs), _vp]),
    ("imh_groupnorm_workspace_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("imh_groupnorm_stats_blocks", C.c_int, [C.c_int, C.c_int]),
    ("imh_groupnorm_stats_sub", C.c_int, [C.c_int, C.c_int]),
    ("imh_layernorm", C.c_int, [C.POINTER(NormArgs), _vp]),
    ("imh_elementwise", C.c_int, [C.c_int, C.POINTER(EwArgs), _vp]),
    ("imh_f32", C.c_int, [C.c_int, C.POINTER(F32Args), _vp]),
    ("imh_plan_create", _vp, []),
    ("imh_plan_destroy", None, [_vp]),
    ("imh_plan_add", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int]),
    ("imh_plan_size", C.c_int, [_vp]),
    ("imh_plan_update", C.c_int, [_vp, C.c_int, _vp]),
    ("imh_plan_run", C.c_int, [_vp, _vp]),
    ("imh_plan_run_range", C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    ("imh_plan_capture", C.c_int, [_vp, _vp]),
    ("imh_plan_replay", C.c_int, [_vp, _vp]),
    ("imh_plan_time_ops", C.c_int, [_vp, _vp, C.POINTER(C.c_float), C.c_int]),
    ("imh_plan_get_tag", C.c_int, [_vp, C.c_int]),
    ("imh_plan_get_kind", C.c_int, [_vp, C.c_int]),
]

_lib = None


class ImhError(RuntimeError):
    pass


def load():
    """Load libimh_hip.so (once).  Raises ImhError if it is not built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImhError(f"{LIB_PATH} is missing: build the HIP extension first "
                       f"(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.imh_abi_version() != ABI_VERSION:
        raise ImhError("libimh_hip.so ABI version mismatch")
    # measurement knob (tools/, alternating bench.py processes): IMH_DEBUG_SET="key=value,..." -> imh_debug_set at load
    for kv in filter(None, os.environ.get("IMH_DEBUG_SET", "").split(",")):
        k, v = kv.split("=")
        lib.imh_debug_set(int(k), int(v))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().imh_last_error()
        raise ImhError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
