"""ctypes binding of libray_amd.so (include/ray_amd.h).  No fallback: if the shared library is
missing or a call fails, an exception is raised -- there is no CPU path in the product."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAY_AMD_LIB") or os.path.join(_HERE, "libray_amd.so")

RAY_OK = 0
ST_EXITED, ST_STEP_CAP, ST_BISECT_CAP, ST_OOB, ST_CEASED = 0x01, 0x02, 0x04, 0x08, 0x10
OB_NONE, OB_CIRCLE, OB_TREFOIL = 0, 1, 2
TIME_SECONDS, TIME_STEPS = 0, 1
TRACE_SHORTEN_CEASE, TRACE_OVERSHOOT = 0, 1
ARITH_EXACT, ARITH_FUSED = 0, 1
STREAM_OWN, STREAM_BORROWED = 0, 1
OPT_SKIP_INIT_COLUMNS = 1
OPT_SOURCE_STRIDE = 2
OPT_FAN_DIRECTIONS = 3
OPT_TUNE_BLOCK_THREADS, OPT_TUNE_WAVE_DIRS, OPT_TUNE_SORT_KEY_BITS = 4, 5, 6
OPT_TUNE_FRECHET_LDS_ROWS, OPT_TUNE_MERGE_GRID, OPT_TUNE_FRECHET_SLAB_MB = 7, 8, 9
OPT_TUNE_MERGE_ENTRIES = 10


class RayError(RuntimeError):
    pass


class Obstacle(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("r0", C.c_double),
                ("reserved", C.c_double * 2)]


class ShootParams(C.Structure):
    _fields_ = [("ns", C.c_int32), ("nd", C.c_int32), ("src_begin", C.c_int32),
                ("src_end", C.c_int32), ("dir_lo", C.c_double), ("dir_hi", C.c_double),
                ("dt", C.c_double), ("time_unit", C.c_int32), ("max_steps", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("ray_steps", C.c_uint64), ("launches", C.c_uint64),
                ("kernel_ms", C.c_double)]


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> argtypes; every entry point include/ray_amd.h declares (tests check the list is complete)
SIGNATURES = {
    "ray_version": [],
    "ray_last_error": [],
    "ray_runtime_info": [C.c_char_p, C.c_size_t],
    "ray_linspace": [_f64, _f64, C.c_int, _vp],
    "ray_device_count": [C.POINTER(C.c_int)],
    "ray_ctx_create": [C.c_int, C.POINTER(_vp)],
    "ray_ctx_destroy": [_vp],
    "ray_ctx_set_stream": [_vp, C.c_int, _vp],
    "ray_ctx_synchronize": [_vp],
    "ray_ctx_release_workspaces": [_vp],
    "ray_ctx_set_arith": [_vp, C.c_int],
    "ray_ctx_set_option": [_vp, C.c_int, C.c_int],
    "ray_host_register": [_vp, C.c_size_t],
    "ray_host_unregister": [_vp],
    "ray_medium_create_grid": [_vp, _vp, C.c_int, _f64, C.POINTER(_vp)],
    "ray_medium_update_grid": [_vp, _vp],
    "ray_medium_update_grid_dev": [_vp, _vp],
    "ray_medium_export_tables": [_vp, _vp, _vp, _vp],
    "ray_medium_create_sinprod": [_vp, _f64, _f64, _f64, _f64, _f64, C.POINTER(_vp)],
    "ray_medium_create_radial": [_vp, _i32, _vp, _vp, _vp, C.POINTER(_vp)],
    "ray_medium_destroy": [_vp],
    "ray_shoot_to_exit": [_vp, _vp, C.POINTER(Obstacle), C.POINTER(ShootParams), _vp, _i64, _vp, _vp, _vp],
    "ray_shoot_to_exit_dev": [_vp, _vp, C.POINTER(Obstacle), C.POINTER(ShootParams), _vp, _i64, _vp, _vp, _vp],
    "ray_shoot_states": [_vp, _vp, C.POINTER(Obstacle), _f64, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp],
    "ray_shoot_states_dev": [_vp, _vp, C.POINTER(Obstacle), _f64, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp],
    "ray_trace_fixed_time": [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _vp, _i64, _vp, _vp],
    "ray_trace_fixed_time_dev": [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _vp, _i64, _vp, _vp],
    "ray_forward_operator": [_vp, _vp, _vp, _i64, _i64, _f64, _f64, _vp, _i64, _vp, _vp, _vp, C.POINTER(_vp)],
    "ray_frechet_shape": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "ray_frechet_export_csr": [_vp, _vp, _vp, _vp],
    "ray_frechet_csr_dev": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "ray_frechet_block_rows_dev": [_vp, C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "ray_frechet_destroy": [_vp],
    "ray_frechet_dofs": [_vp, _vp, _vp],
    "ray_frechet_dofs_dev": [_vp, _vp, _vp],
    "ray_frechet_matvec": [_vp, _vp, _vp],
    "ray_frechet_matvec_dev": [_vp, _vp, _vp],
    "ray_frechet_matvec_t": [_vp, _vp, _vp, _vp],
    "ray_frechet_matvec_t_dev": [_vp, _vp, _vp, _vp],
    "ray_frechet_normal_matvec": [_vp, _vp, _vp, _vp],
    "ray_frechet_normal_matvec_dev": [_vp, _vp, _vp, _vp],
    "ray_frechet_normal_diag": [_vp, _vp, _vp],
    "ray_frechet_normal_diag_dev": [_vp, _vp, _vp],
    "ray_frechet_fidelity_update": [_vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp],
    "ray_frechet_fidelity_update_dev": [_vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp],
    "ray_trace_paths": [_vp, _vp, _vp, _i64, _i64, _f64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp],
    "ray_shoot_paths": [_vp, _vp, C.POINTER(Obstacle), _f64, _i32, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "ray_classify": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "ray_classify_dev": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    "ray_misfit": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _f64, _vp],
    "ray_misfit_dev": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _f64, _vp],
    "ray_group_create": [C.POINTER(C.c_int), C.c_int, C.POINTER(_vp)],
    "ray_group_destroy": [_vp],
    "ray_group_size": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "ray_group_set_arith": [_vp, C.c_int],
    "ray_group_medium_create_grid": [_vp, _vp, C.c_int, _f64, C.POINTER(_vp)],
    "ray_group_medium_create_sinprod": [_vp, _f64, _f64, _f64, _f64, _f64, C.POINTER(_vp)],
    "ray_group_medium_update_grid": [_vp, _vp],
    "ray_group_medium_destroy": [_vp],
    "ray_group_shoot_to_exit": [_vp, _vp, C.POINTER(Obstacle), C.POINTER(ShootParams), _vp, _i64, _vp, _vp, _vp],
    "ray_group_trace_fixed_time": [_vp, _vp, _vp, _i64, _i64, C.c_int32, _f64, C.c_int32, _vp, _i64, _vp, _vp],
    "ray_group_forward_operator": [_vp, _vp, _vp, _i64, _i64, C.c_int32, _f64, _f64, _vp, _i64, _vp, _vp, C.POINTER(_vp)],
    "ray_group_frechet_shape": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "ray_group_frechet_member": [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(_i64)],
    "ray_group_frechet_export_csr": [_vp, _vp, _vp, _vp],
    "ray_group_frechet_matvec_t": [_vp, _vp, _vp, _vp],
    "ray_group_frechet_normal_matvec": [_vp, _vp, _vp, _vp],
    "ray_group_frechet_destroy": [_vp],
    "ray_group_set_observation": [_vp, _vp, _i64, _i64],
    "ray_group_misfit": [_vp, _vp, _vp, _f64, _vp],
    "ray_group_get_counters": [_vp, C.POINTER(Counters)],
    "ray_misfit_static_dev": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _f64, _vp],
    "ray_ctx_enable_timing": [_vp, C.c_int],
    "ray_get_counters": [_vp, C.POINTER(Counters)],
    "ray_reset_counters": [_vp],
}

_lib = None


def load():
    """Load libray_amd.so (built in-tree by `make -C ray_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RayError(f"{LIB_PATH} not found: build it with `make -C ray_amd/csrc` "
                       "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "ray_last_error" else C.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != RAY_OK:
        msg = load().ray_last_error()
        raise RayError(f"libray_amd error {rc}: {msg.decode() if msg else '?'}")
