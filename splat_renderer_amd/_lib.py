"""ctypes loader for libsplat_hip.so (the C ABI in include/splat.h).

There is no CPU fallback: if the library is missing or no gfx950 device is visible the import of
the library / creation of a context raises.  This module never imports anything from oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPLAT_LIB_PATH: another build of the library (developer hook for same-box A/B runs of two kernel versions)
LIB_PATH = os.environ.get("SPLAT_LIB_PATH") or os.path.join(_HERE, "libsplat_hip.so")

OK = 0
ERR_NAMES = {-1: "INVALID", -2: "HIP", -3: "OOM", -4: "CAPACITY", -5: "STATE", -6: "NO_DEVICE", -7: "COMM", -8: "RETRY"}
# "the previous frame must be rendered again" (its sync-free pair limit overflowed / its lists failed the order check)
ERR_CAPACITY, ERR_RETRY = -4, -8
RENDER_AGAIN = (ERR_CAPACITY, ERR_RETRY)  # reports about the PREVIOUS sync-free frame: overflowed its pair limit / failed the order check

STAGE_PROJECT, STAGE_SORT, STAGE_BIN, STAGE_COMPOSITE, STAGE_EXCHANGE, STAGE_BIN_SCATTER, STAGE_BIN_PASS2, STAGE_BIN_TILE_SORT = range(8)
STAGE_NAMES = ("project", "sort", "bin", "composite", "exchange", "bin_scatter", "bin_second_pass", "bin_tile_sort")
TIMING_COUNT_ENTRIES = 0x80000000  # splat_set_timing_stages: also count the entries a timed frame's composite staged / consumed
MODE_FRONT_TO_BACK, MODE_REFERENCE_LITERAL = 0, 1
RECORDS_PROJECTED, RECORDS_COMPACT, RECORDS_DISC48, RECORDS_LIT32 = 0, 1, 2, 3
FOOTPRINT_ISOTROPIC, FOOTPRINT_DISC = 0, 1
U32_MAX = 0xFFFFFFFF


class SplatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsplat_hip error {ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


SDF_SPHERE, SDF_BOX, SDF_TORUS, SDF_CAPSULE, SDF_UNION, SDF_INTERSECTION, SDF_SUBTRACTION, SDF_SMOOTH_UNION = 0, 1, 2, 3, 16, 17, 18, 19
SDF_MAX_INSTR = 32


class SdfInstr(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", C.c_float * 7)]


class CompositeCfg(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("early_out", C.c_uint32), ("tile_size", C.c_uint32),
                ("tile_row0", C.c_uint32), ("tile_row1", C.c_uint32), ("record_format", C.c_uint32),
                ("prelit", C.c_uint32), ("footprint", C.c_uint32)]


# name -> (restype, argtypes); the single source of truth checked against include/splat.h by
# tests/test_abi_cpu.py
_vp, _u32, _sz, _i = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
_pvp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "splat_abi_version": (_i, []),
    "splat_ctx_create": (_i, [_i, _pvp]),
    "splat_ctx_create_on_stream": (_i, [_i, _vp, _pvp]),
    "splat_ctx_destroy": (None, [_vp]),
    "splat_last_error": (C.c_char_p, [_vp]),
    "splat_sync": (_i, [_vp]),
    "splat_set_timing": (_i, [_vp, _i]),
    "splat_set_timing_stages": (_i, [_vp, _u32]),
    "splat_set_timing_sampling": (_i, [_vp, _u32]),
    "splat_stage_time_ms": (_i, [_vp, _i, C.POINTER(C.c_float)]),
    "splat_stage_time_stats": (_i, [_vp, _i, C.POINTER(_u32), C.POINTER(C.c_double)]),
    "splat_timing_consumed": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "splat_buf_alloc": (_i, [_vp, _sz, _pvp]),
    "splat_buf_free": (_i, [_vp, _vp]),
    "splat_buf_upload": (_i, [_vp, _vp, _vp, _sz]),
    "splat_buf_download": (_i, [_vp, _vp, _vp, _sz]),
    "splat_buf_zero": (_i, [_vp, _vp, _sz]),
    "splat_buf_copy": (_i, [_vp, _vp, _vp, _sz]),
    "splat_update_props": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "splat_update_props_planes": (_i, [_vp, _vp, _vp, _u32, _vp, _vp]),
    "splat_props_to_planes": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "splat_project": (_i, [_vp, C.POINTER(C.c_float), _vp, _u32, _u32, _vp, _vp, _vp, _u32]),
    "splat_project_disc": (_i, [_vp, C.POINTER(C.c_float), _vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _u32]),
    "splat_extract_keys": (_i, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "splat_sort_create": (_i, [_vp, _u32, _pvp]),
    "splat_sort_destroy": (None, [_vp]),
    "splat_sort_capacity": (_u32, [_vp]),
    "splat_sort_keys": (_vp, [_vp]),
    "splat_sort_payload": (_vp, [_vp]),
    "splat_sort_run": (_i, [_vp, _u32, _u32, _u32]),
    "splat_sort_sorted_payload": (_vp, [_vp]),
    "splat_sort_sorted_keys": (_vp, [_vp]),
    "splat_probe_lds_atomic_order": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "splat_sort_set_mode": (_i, [_vp, _i]),
    "splat_rank_status": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_u32)]),
    "splat_composite_forget_history": (_i, [_vp]),
    "splat_composite_options": (_i, [_vp, _i, _i, _i]),
    "splat_scan_u32": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "splat_bin_create": (_i, [_vp, _u32, _pvp]),
    "splat_bin_destroy": (None, [_vp]),
    "splat_bin_run": (_i, [_vp, _vp, _u32, _vp, _u32, _u32, _u32, _u32, _u32]),
    "splat_bin_tile_size": (_u32, [_vp]),
    "splat_bin_counts": (_i, [_vp, _pvp]),
    "splat_bin_offsets": (_i, [_vp, _pvp]),
    "splat_bin_indices": (_i, [_vp, _pvp]),
    "splat_bin_total": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "splat_bin_set_frame_order": (_i, [_vp, _i]),
    "splat_bin_dims": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "splat_validate_tile_order": (_i, [_vp, _vp, _vp, _u32, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "splat_lit_colors": (_i, [_vp, _vp, _u32, _vp, _u32, _u32, _vp]),
    "splat_composite": (_i, [_vp, C.POINTER(CompositeCfg), _vp, _u32, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _u32,
                             _vp, _vp, _vp]),
    "splat_render_frame": (_i, [_vp, _vp, _vp, C.POINTER(CompositeCfg), C.POINTER(C.c_float), _vp, _vp, _u32, _u32,
                                _u32, _vp, _vp, _vp]),
    "splat_render_frame_planes": (_i, [_vp, _vp, _vp, C.POINTER(CompositeCfg), C.POINTER(C.c_float), _vp, _vp, _vp, _u32,
                                       _u32, _u32, _vp, _vp, _vp]),
    "splat_project_slice": (_i, [_vp, C.POINTER(C.c_float), _vp, _u32, _u32, _u32, _vp]),
    "splat_project_slice_compact": (_i, [_vp, C.POINTER(C.c_float), _vp, _u32, _u32, _u32, _vp]),
    "splat_project_slice_disc": (_i, [_vp, C.POINTER(C.c_float), _vp, _u32, _vp, _u32, _u32, _u32, _vp]),
    "splat_expand_compact": (_i, [_vp, _vp, _u32, _u32, _vp]),
    "splat_band_frame": (_i, [_vp, _vp, _vp, C.POINTER(CompositeCfg), _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "splat_band_settle": (_i, [_vp, _vp, _vp, C.POINTER(_u32), C.POINTER(C.c_uint64)]),
    "splat_band_kept": (_i, [_vp, _vp, C.POINTER(_u32)]),
    "splat_band_keys": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, C.POINTER(_u32)]),
    "splat_sdf_gradients": (_i, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "splat_sdf_update_positions": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "splat_sdf_scale_factors": (_i, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "splat_sdf_curvature": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "splat_sdf_seed_positions": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), _u32, C.c_uint64, _vp]),
    "splat_sdf_generate": (_i, [_vp, _vp, _u32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint64, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "splat_comm_unique_id": (_i, [_vp]),
    "splat_comm_init": (_i, [_vp, _i, _i, _vp, _pvp]),
    "splat_comm_destroy": (None, [_vp]),
    "splat_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "splat_comm_count": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "splat_allgather_records": (_i, [_vp, _vp, _vp, _vp, _sz]),
}
COMM_ID_BYTES = 128
# include/splat.h, #ifdef SPLAT_TEST_HOOKS: exported by the test build only (libsplat_hip_hooks.so, which tests/ and tools/
# select with SPLAT_LIB_PATH); bound when the loaded library has them
HOOK_SIGNATURES = {
    "splat_debug_inject_order_fault": (_i, [_vp, _u32, _u32]),
    "splat_debug_set_tile_order": (_i, [_vp, _vp]),
    "splat_debug_set_tile_sort_order": (_i, [_vp, _vp]),
    "splat_debug_rerun_tile_sort": (_i, [_vp, _vp]),
    "splat_debug_tile_sort_launches": (_i, [_vp, C.POINTER(_u32)]),
    "splat_debug_lds_rate": (_i, [_vp, _i, _u32, _u32, C.POINTER(C.c_float)]),
}
HOOKS_LIB_PATH = os.path.join(_HERE, "libsplat_hip_hooks.so")

_lib = None


class _Bound:
    """Only the functions SIGNATURES declares, with their argument types bound.  (A raw CDLL hands out any
    exported symbol with C's default int arguments, which truncates 64-bit device pointers: calling a
    function that was added to splat.h but not to SIGNATURES must fail here, not fault on the GPU.)"""

    def __getattr__(self, name):
        if name in HOOK_SIGNATURES:
            raise AttributeError(f"libsplat_hip: {name} is a test hook: the shipped library does not carry it; run with "
                                 f"SPLAT_LIB_PATH={HOOKS_LIB_PATH} (the -DSPLAT_TEST_HOOKS build)")
        raise AttributeError(f"libsplat_hip: {name} is not declared in splat_renderer_amd._lib.SIGNATURES")


def load():
    """Load libsplat_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). splat_renderer_amd has no CPU fallback.")
    dll = C.CDLL(LIB_PATH)
    lib = _Bound()
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(dll, name)  # AttributeError here = the .so does not export what splat.h declares
        fn.restype = res
        fn.argtypes = args
        setattr(lib, name, fn)
    lib.has_hooks = hasattr(dll, "splat_debug_inject_order_fault")
    if lib.has_hooks:
        for name, (res, args) in HOOK_SIGNATURES.items():
            fn = getattr(dll, name)
            fn.restype = res
            fn.argtypes = args
            setattr(lib, name, fn)
    lib._dll = dll
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != OK:
        msg = load().splat_last_error(ctx)
        raise SplatError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
