"""ctypes description of include/rtgr.h and the loader of librtgr_hip.so.

The structures mirror include/rtgr.h field for field; the loader FAILS LOUDLY when the HIP library is missing —
there is no CPU fallback in the product (the CPU oracle under oracle/ is test infrastructure and is never loaded
from here).
"""
import ctypes as C
import os

RTGR_MAX_OBJECTS = 16          # objects held inline in rtgr_scene.obj; a longer list goes through rtgr_scene.objects
RTGR_OBJECTS_LIMIT = 1 << 20
RTGR_ABI_VERSION = 4
RTGR_MAX_DEVICES = 16
RTGR_MAX_SOURCES = 16

# enum rtgr_metric
MINKOWSKI, KS_REF, KS_TRUE, USER = 0, 1, 2, 3
METRIC_GENERIC = 0x100  # RTGR_METRIC_GENERIC flag
# enum rtgr_object_kind
PLANE, SPHERE, DISK, USER_OBJECT = 1, 2, 3, 4
# enum rtgr_ray_status
RAY_EVENT, RAY_LAMBDA1, RAY_MAXSTEPS, RAY_DTMIN, RAY_NAN = 0, 1, 2, 3, 4
# enum rtgr_status
OK, ERR_BAD_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NAN_INPUT, ERR_NOT_INIT = 0, -1, -2, -3, -4, -5


class rtgr_object(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("type", C.c_uint32), ("p", C.c_double * 9)]


class rtgr_scene(C.Structure):
    _fields_ = [("metric", C.c_uint32), ("nobj", C.c_uint32), ("M", C.c_double), ("a", C.c_double),
                ("user_metric", C.c_uint64), ("obj", rtgr_object * RTGR_MAX_OBJECTS), ("objects", C.POINTER(rtgr_object))]

    def object(self, o):
        """object o of the list, wherever it lives (the inline slots, or the caller array `objects`)"""
        return self.objects[o] if self.objects else self.obj[o]

    def clone(self):
        """a copy that keeps the list of a long scene alive with it (the `objects` array is a Python object of its own)"""
        c = type(self).from_buffer_copy(self)
        if hasattr(self, "_keep"):
            c._keep = self._keep
        return c


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


class rtgr_unit_info(C.Structure):
    _fields_ = [("metric", C.c_uint32), ("spin", C.c_uint32), ("has_objects", C.c_uint32), ("has_reach", C.c_uint32),
                ("far_waves", C.c_uint32), ("near_waves", C.c_uint32), ("f32_waves", C.c_uint32), ("probe_ok", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class rtgr_ray_outputs(C.Structure):
    _fields_ = [("state_end", C.c_void_p), ("lambda_end", C.c_void_p), ("status", C.c_void_p),
                ("hit", C.c_void_p), ("n_accept", C.c_void_p), ("n_reject", C.c_void_p), ("redshift", C.c_void_p),
                ("hit32", C.c_void_p)]


# Every symbol include/rtgr.h declares (tests check the .so exports exactly this list).
EXPORTS = [
    "rtgr_create", "rtgr_destroy", "rtgr_context_devices", "rtgr_trim", "rtgr_init", "rtgr_shutdown", "rtgr_last_error",
    "rtgr_abi_version", "rtgr_solver_defaults", "rtgr_device_info", "rtgr_set_option", "rtgr_get_option",
    "rtgr_reserve_workspace", "rtgr_timing_enable", "rtgr_timing_read", "rtgr_timing_read_exchange", "rtgr_peer_access",
    "rtgr_trace_device_f64", "rtgr_trace_device_f32", "rtgr_trace_rows_device_f64", "rtgr_trace_rows_device_f32",
    "rtgr_trace_f64", "rtgr_trace_f32", "rtgr_trace_pixels_f64", "rtgr_trace_pixels_f32", "rtgr_trace_one_f64", "rtgr_trace_one_f32",
    "rtgr_trace_sharded_f64", "rtgr_trace_sharded_device_f64", "rtgr_trace_sharded_f32", "rtgr_trace_sharded_device_f32",
    "rtgr_make_canvas_device_f64", "rtgr_make_canvas_f64", "rtgr_make_canvas_device_f32", "rtgr_make_canvas_f32",
    "rtgr_eval_metric_f64", "rtgr_eval_metric_f32", "rtgr_eval_geodesic_f64", "rtgr_eval_geodesic_f32",
    "rtgr_eval_fastmath_f64", "rtgr_quantize_device_f64",
    "rtgr_user_metric_load", "rtgr_user_metric_compile", "rtgr_user_metric_unload", "rtgr_user_metric_loaded", "rtgr_code_object_audit", "rtgr_user_metric_build", "rtgr_listing_repair",
    "rtgr_user_unit_compile", "rtgr_user_unit_build", "rtgr_user_unit_info", "rtgr_scene_check",
    "rtgr_eval_objects_f64", "rtgr_eval_objects_f32", "rtgr_user_source_join",
    "rtgr_trace_frames_f64", "rtgr_trace_frames_f32", "rtgr_trace_frames_pixels_f64", "rtgr_trace_frames_pixels_f32",
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
    """argtypes of every entry point.  The first argument of every compute entry point is the rtgr_context* (None = the
    process's default context)."""
    vp, u64, i32, clong = C.c_void_p, C.c_uint64, C.c_int, C.c_long
    P = C.POINTER
    ctx = vp
    lib.rtgr_create.argtypes = [P(i32), i32, P(vp)]
    lib.rtgr_destroy.argtypes = [ctx]
    lib.rtgr_context_devices.argtypes = [ctx]
    lib.rtgr_trim.argtypes = [ctx]
    lib.rtgr_init.argtypes = [i32]
    lib.rtgr_shutdown.argtypes = []
    lib.rtgr_last_error.argtypes = []
    lib.rtgr_last_error.restype = C.c_char_p
    lib.rtgr_abi_version.argtypes = []
    lib.rtgr_solver_defaults.argtypes = [P(rtgr_solver), i32]
    lib.rtgr_device_info.argtypes = [ctx, i32, C.c_char_p, u64, P(i32), P(i32), P(i32)]
    lib.rtgr_set_option.argtypes = [ctx, C.c_char_p, clong]
    lib.rtgr_get_option.argtypes = [ctx, C.c_char_p, P(clong)]
    lib.rtgr_reserve_workspace.argtypes = [ctx, vp, vp, u64, i32, i32]
    lib.rtgr_timing_enable.argtypes = [ctx, i32, i32]
    lib.rtgr_timing_read.argtypes = [ctx, i32, P(C.c_double * 4), P(C.c_uint64 * 4)]
    lib.rtgr_timing_read_exchange.argtypes = [ctx, i32, P(C.c_double * 2), P(C.c_uint64 * 2)]
    lib.rtgr_peer_access.argtypes = [ctx, i32, C.c_char_p, u64]
    for suf in ("f64", "f32"):
        getattr(lib, f"rtgr_trace_device_{suf}").argtypes = [
            ctx, P(rtgr_scene), P(rtgr_solver), vp, P(rtgr_camera), u64, u64, u64, u64, vp, P(rtgr_ray_outputs), vp, vp]
        getattr(lib, f"rtgr_trace_rows_device_{suf}").argtypes = [
            ctx, P(rtgr_scene), P(rtgr_solver), P(rtgr_camera), u64, u64, u64, u64, u64, vp, P(rtgr_ray_outputs), vp, vp]
        getattr(lib, f"rtgr_trace_{suf}").argtypes = [
            ctx, P(rtgr_scene), P(rtgr_solver), vp, P(rtgr_camera), u64, u64, u64, u64, vp, P(rtgr_ray_outputs),
            P(rtgr_counters)]
        getattr(lib, f"rtgr_eval_metric_{suf}").argtypes = [ctx, P(rtgr_scene), vp, u64, vp, vp, vp]
        getattr(lib, f"rtgr_eval_geodesic_{suf}").argtypes = [ctx, P(rtgr_scene), vp, u64, i32, vp]
    for f in (lib.rtgr_trace_pixels_f64, lib.rtgr_trace_pixels_f32):
        f.argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), vp, u64, u64, vp, P(rtgr_counters)]
    for f in (lib.rtgr_trace_one_f64, lib.rtgr_trace_one_f32):
        f.argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), vp, vp, vp, vp, vp]
    for f in (lib.rtgr_trace_frames_f64, lib.rtgr_trace_frames_f32):     # (arrays of per-frame pointers: c_void_p arrays)
        f.argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), C.c_uint32, P(rtgr_camera), P(vp), u64, u64, P(vp), P(rtgr_ray_outputs), P(rtgr_counters)]
    for f in (lib.rtgr_trace_frames_pixels_f64, lib.rtgr_trace_frames_pixels_f32):
        f.argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), C.c_uint32, P(vp), u64, u64, P(vp), P(rtgr_counters)]
    for name in ("rtgr_trace_sharded_f64", "rtgr_trace_sharded_device_f64", "rtgr_trace_sharded_f32", "rtgr_trace_sharded_device_f32"):
        getattr(lib, name).argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), P(rtgr_camera), u64, u64, vp,
                                       P(rtgr_ray_outputs), P(rtgr_counters)]
    for suf in ("f64", "f32"):
        getattr(lib, f"rtgr_make_canvas_device_{suf}").argtypes = [ctx, P(rtgr_scene), P(rtgr_camera), u64, u64, u64, u64, vp, vp]
        getattr(lib, f"rtgr_make_canvas_{suf}").argtypes = [ctx, P(rtgr_scene), P(rtgr_camera), u64, u64, u64, u64, vp]
    lib.rtgr_eval_fastmath_f64.argtypes = [ctx, vp, u64, vp, vp]
    lib.rtgr_quantize_device_f64.argtypes = [ctx, vp, u64, u64, vp, vp]
    lib.rtgr_user_metric_load.argtypes = [ctx, C.c_char_p, P(u64)]
    lib.rtgr_user_metric_compile.argtypes = [ctx, C.c_char_p, i32, P(u64)]
    lib.rtgr_user_metric_unload.argtypes = [ctx, u64]
    lib.rtgr_user_metric_loaded.argtypes = [ctx, u64]
    lib.rtgr_code_object_audit.argtypes = [C.c_char_p, P(i32), C.c_char_p, u64]
    lib.rtgr_user_metric_build.argtypes = [C.c_char_p, i32, C.c_char_p]
    lib.rtgr_listing_repair.argtypes = [C.c_char_p, C.c_char_p, P(i32)]
    lib.rtgr_user_unit_compile.argtypes = [ctx, C.c_char_p, i32, P(rtgr_scene), P(u64)]
    lib.rtgr_user_unit_build.argtypes = [C.c_char_p, i32, P(rtgr_scene), C.c_char_p]
    lib.rtgr_user_unit_info.argtypes = [ctx, u64, P(rtgr_unit_info)]
    lib.rtgr_user_source_join.argtypes = [P(C.c_char_p), P(C.c_uint32), i32, C.c_char_p, u64, P(u64)]
    for suf in ("f64", "f32"):
        getattr(lib, f"rtgr_eval_objects_{suf}").argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), vp, u64, vp, vp, vp, vp]
    lib.rtgr_scene_check.argtypes = [ctx, P(rtgr_scene), P(rtgr_solver), P(rtgr_camera), u64, u64, i32]
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


class options:
    """`with options(lib, split=0, near_early=8): ...` — set launch-policy options of a context (default: the
    process's default context) for the duration of the block (rtgr_set_option).  Scheduling options never change a result
    bit; `tile` and `pack` select another formulation of the same algorithm (equal up to rounding): include/rtgr.h."""

    def __init__(self, lib, ctx=None, **kw):
        self.lib, self.ctx, self.kw, self.old = lib, ctx, kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            cur = C.c_long(0)
            check(self.lib, self.lib.rtgr_get_option(self.ctx, k.encode(), C.byref(cur)))
            self.old[k] = cur.value
            check(self.lib, self.lib.rtgr_set_option(self.ctx, k.encode(), int(v)))
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            check(self.lib, self.lib.rtgr_set_option(self.ctx, k.encode(), v))
        return False


def create_context(lib, device_ids):
    """rtgr_create over a list of HIP device ordinals -> opaque context handle (pass it as the first argument)."""
    ids = (C.c_int * len(device_ids))(*device_ids)
    h = C.c_void_p(None)
    check(lib, lib.rtgr_create(ids, len(device_ids), C.byref(h)))
    return h
