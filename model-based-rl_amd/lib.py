"""ctypes binding of the C ABI in include/mbrl.h (csrc/libmbrl_hip.so).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, loading /
model creation raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MBRL_LIB") or os.path.join(HERE, "csrc", "libmbrl_hip.so")   # MBRL_LIB: profiling variants

ST_NONFINITE, ST_NOCONVERGE, ST_OVERFLOW, ST_DEGENERATE, ST_BADPLAN = 1, 2, 4, 8, 16
FD_FORWARD, FD_CENTRAL = 0, 1
STATIC_MAX_UNITS = 32768
EPISODE_MAJOR, TIME_MAJOR = 0, 1

SYMBOLS = ("mbrl_abi_version", "mbrl_model_create", "mbrl_model_destroy", "mbrl_model_dims", "mbrl_step",
           "mbrl_fd_jacobians", "mbrl_fd_jacobians_closed_loop", "mbrl_fd_plan_bytes", "mbrl_fd_plan",
           "mbrl_fd_jacobians_planned", "mbrl_fd_jacobians_static", "mbrl_step_record", "mbrl_rollout_record", "mbrl_fd_plan_finish", "mbrl_vjp_scan", "mbrl_vjp_scan_layout", "mbrl_vjp_scan_sharded", "mbrl_lane_map", "mbrl_last_error")

_lib = None
vp = C.c_void_p


class MbrlError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MbrlError("HIP extension not built: %s is missing (run __graft_entry__.build()); "
                        "this package has no CPU fallback" % LIB_PATH)
    # torch ships its own HIP runtime (libamdhip64): load it FIRST so that this library binds to the
    # same runtime instance as torch's streams/allocations instead of a second copy from /opt/rocm.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.mbrl_abi_version.restype = C.c_int
    L.mbrl_last_error.restype = C.c_char_p
    L.mbrl_model_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.mbrl_model_destroy.argtypes = [vp]
    L.mbrl_model_destroy.restype = None
    L.mbrl_model_dims.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_double)]
    L.mbrl_step.argtypes = [vp, C.c_int64] + [vp] * 7 + [vp]
    L.mbrl_fd_jacobians.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_double, C.c_int, C.c_int] + [vp] * 7 + [vp]
    L.mbrl_fd_jacobians_closed_loop.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_double, C.c_int, C.c_int] + [vp] * 7 + [vp]
    L.mbrl_fd_plan_bytes.argtypes = [vp, C.c_int64]
    L.mbrl_fd_plan_bytes.restype = C.c_int64
    L.mbrl_fd_plan.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_int, vp, C.c_int64, vp, vp, vp, vp]
    L.mbrl_fd_jacobians_planned.argtypes = [vp, C.c_int64, vp, vp, vp, vp, C.c_double, C.c_int, C.c_int, vp, C.c_int64] + \
                                           [vp] * 5 + [vp]
    L.mbrl_fd_jacobians_static.argtypes = L.mbrl_fd_jacobians.argtypes
    L.mbrl_vjp_scan.argtypes = [vp, C.c_int64, C.c_int, C.c_int] + [vp] * 7 + [vp]
    L.mbrl_vjp_scan_layout.argtypes = [vp, C.c_int64, C.c_int, C.c_int, C.c_int] + [vp] * 7 + [vp]
    L.mbrl_vjp_scan_sharded.argtypes = [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64] + [vp] * 7 + [vp]
    L.mbrl_step_record.argtypes = [vp, C.c_int64] + [vp] * 7 + [vp, C.c_int64, C.c_int64, C.c_int64, vp]
    L.mbrl_fd_plan_finish.argtypes = [vp, C.c_int64, vp, vp, C.c_int64, vp]
    L.mbrl_rollout_record.argtypes = [vp, C.c_int64, C.c_int] + [vp] * 5 + [vp, C.c_int64, C.c_int64, C.c_int64, vp]
    # test knobs outside include/mbrl.h (tests/test_gpu_overflow.py, tests/test_zz_no_alloc.py)
    L.mbrl_debug_set_flags.argtypes = [C.c_int]
    L.mbrl_debug_static_hold.argtypes = [vp, C.c_int, vp]
    L.mbrl_stream_probe_spin.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]     # engine.py: side-stream selection
    if L.mbrl_abi_version() != 9:
        raise MbrlError("ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise MbrlError("%s (code %d)" % (load().mbrl_last_error().decode(), rc))
