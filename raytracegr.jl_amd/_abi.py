"""ctypes description of include/rtgr.h and the loader of librtgr_hip.so.

The structures mirror include/rtgr.h field for field; the loader FAILS LOUDLY when the HIP library is missing —
there is no CPU fallback in the product (the CPU oracle under oracle/ is test infrastructure and is never loaded
from here).
"""
import ctypes as C
import os

RTGR_MAX_OBJECTS = 16
RTGR_ABI_VERSION = 1

# enum rtgr_metric
MINKOWSKI, KS_REF, KS_TRUE, USER = 0, 1, 2, 3
METRIC_GENERIC = 0x100  # RTGR_METRIC_GENERIC flag
# enum rtgr_object_kind
PLANE, SPHERE, DISK = 1, 2, 3
# enum rtgr_ray_status
RAY_EVENT, RAY_LAMBDA1, RAY_MAXSTEPS, RAY_DTMIN, RAY_NAN = 0, 1, 2, 3, 4
# enum rtgr_status
OK, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NAN_INPUT, ERR_NOT_INIT = 0, -1, -2, -3, -4, -5


class rtgr_object(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("p", C.c_double * 9)]


class rtgr_scene(C.Structure):
    _fields_ = [("metric", C.c_uint32), ("nobj", C.c_uint32), ("M", C.c_double), ("a", C.c_double),
                ("obj", rtgr_object * RTGR_MAX_OBJECTS)]


class rtgr_solver(C.Structure):
    _fields_ = [("reltol", C.c_double), ("abstol", C.c_double), ("lambda0", C.c_double), ("lambda1", C.c_double),
                ("hit_threshold", C.c_double), ("miss_rgb", C.c_double * 3), ("max_steps", C.c_uint32),
                ("interp_points", C.c_uint32)]


class rtgr_camera(C.Structure):
    _fields_ = [("pos", C.c_double * 4), ("widthx", C.c_double * 4), ("widthy", C.c_double * 4),
                ("normal", C.c_double * 4)]


class rtgr_counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("accepted", C.c_uint64), ("rejected", C.c_uint64),
                ("rhs_evals", C.c_uint64), ("events", C.c_uint64), ("events_interior", C.c_uint64),
                ("not_finished", C.c_uint64), ("reserved", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


class rtgr_ray_outputs(C.Structure):
    _fields_ = [("state_end", C.c_void_p), ("lambda_end", C.c_void_p), ("status", C.c_void_p),
                ("hit", C.c_void_p), ("n_accept", C.c_void_p), ("n_reject", C.c_void_p)]


# Every symbol include/rtgr.h declares (tests check the .so exports exactly this list).
EXPORTS = [
    "rtgr_init", "rtgr_shutdown", "rtgr_last_error", "rtgr_abi_version", "rtgr_solver_defaults",
    "rtgr_device_info", "rtgr_reserve_workspace", "rtgr_timing_enable", "rtgr_timing_read", "rtgr_trace_device_f64", "rtgr_trace_device_f32", "rtgr_trace_rows_device_f64", "rtgr_trace_rows_device_f32", "rtgr_trace_f64", "rtgr_trace_f32",
    "rtgr_trace_pixels_f64", "rtgr_trace_one_f64", "rtgr_make_canvas_device_f64", "rtgr_make_canvas_f64",
    "rtgr_eval_metric_f64", "rtgr_eval_geodesic_f64", "rtgr_quantize_device_f64",
    "rtgr_user_metric_load", "rtgr_user_metric_unload", "rtgr_user_metric_loaded",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtgr_hip.so")
_lib = None


class RtgrError(RuntimeError):
    """A negative return code from librtgr_hip.so (message from rtgr_last_error())."""

    def __init__(self, code, msg):
        super().__init__(f"librtgr_hip: error {code}: {msg}")
        self.code = code


def _declare(lib):
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    P = C.POINTER
    lib.rtgr_init.argtypes = [i32]
    lib.rtgr_shutdown.argtypes = []
    lib.rtgr_last_error.argtypes = []
    lib.rtgr_last_error.restype = C.c_char_p
    lib.rtgr_abi_version.argtypes = []
    lib.rtgr_solver_defaults.argtypes = [P(rtgr_solver), i32]
    lib.rtgr_device_info.argtypes = [C.c_char_p, u64, P(i32), P(i32), P(i32)]
    lib.rtgr_reserve_workspace.argtypes = [u64, i32, i32]
    lib.rtgr_timing_enable.argtypes = [i32]
    lib.rtgr_timing_read.argtypes = [P(C.c_double * 4), P(C.c_uint64 * 4)]
    for suf in ("f64", "f32"):
        getattr(lib, f"rtgr_trace_device_{suf}").argtypes = [
            P(rtgr_scene), P(rtgr_solver), vp, P(rtgr_camera), u64, u64, u64, u64, vp, P(rtgr_ray_outputs), vp, vp]
        getattr(lib, f"rtgr_trace_rows_device_{suf}").argtypes = [
            P(rtgr_scene), P(rtgr_solver), P(rtgr_camera), u64, u64, u64, u64, u64, vp, P(rtgr_ray_outputs), vp, vp]
        getattr(lib, f"rtgr_trace_{suf}").argtypes = [
            P(rtgr_scene), P(rtgr_solver), vp, P(rtgr_camera), u64, u64, u64, u64, vp, P(rtgr_ray_outputs),
            P(rtgr_counters)]
    lib.rtgr_trace_pixels_f64.argtypes = [P(rtgr_scene), P(rtgr_solver), vp, u64, u64, vp, P(rtgr_counters)]
    lib.rtgr_trace_one_f64.argtypes = [P(rtgr_scene), P(rtgr_solver), vp, vp, vp, vp, vp]
    lib.rtgr_make_canvas_device_f64.argtypes = [P(rtgr_scene), P(rtgr_camera), u64, u64, u64, u64, vp, vp]
    lib.rtgr_make_canvas_f64.argtypes = [P(rtgr_scene), P(rtgr_camera), u64, u64, u64, u64, vp]
    lib.rtgr_eval_metric_f64.argtypes = [P(rtgr_scene), vp, u64, vp, vp, vp]
    lib.rtgr_eval_geodesic_f64.argtypes = [P(rtgr_scene), vp, u64, i32, vp]
    lib.rtgr_quantize_device_f64.argtypes = [vp, u64, u64, vp, vp]
    lib.rtgr_user_metric_load.argtypes = [C.c_char_p]
    lib.rtgr_user_metric_unload.argtypes = []
    lib.rtgr_user_metric_loaded.argtypes = []
    for name in EXPORTS:
        if name != "rtgr_last_error":
            getattr(lib, name).restype = i32


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7)
    next to libtorch; librtgr_hip.so needs `libamdhip64.so.7` too.  If ours resolved to /opt/rocm's copy first, a later
    `import torch` would bring a second runtime into the process and find "No HIP GPUs".  Loading torch's copy first
    (by path, without importing torch) makes both bind to the same runtime, whichever is used first."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load(path=None):
    """Load librtgr_hip.so (built by __graft_entry__.build() / raytracegr.jl_amd/build.py).

    Raises (never falls back) when the library is missing: the product path is the HIP path.
    """
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RTGR_LIB") or LIB_PATH  # RTGR_LIB: experiment builds (bench A/B only)
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(p)
    _declare(lib)
    got = lib.rtgr_abi_version()
    if got != RTGR_ABI_VERSION:
        raise RuntimeError(f"librtgr_hip ABI version {got} != expected {RTGR_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib, code):
    if code < 0:
        msg = lib.rtgr_last_error()
        raise RtgrError(code, msg.decode() if msg else "?")
    return code
