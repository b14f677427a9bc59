"""Host-side mirror of RayTraceGR.jl's interface for the hot path, over the C ABI of include/rtgr.h.

The reference's host language is Julia, which is absent from this image; this module plays the role of the thin
Julia `ccall` layer (julia/RayTraceGRHIP.jl ships the actual Julia stub): same names, argument meaning and error
behaviour as the reference's exported API, so tests read like the reference's own.

    reference (src/RayTraceGR.jl)                       here
    ------------------------------------------------   --------------------------------------------------
    minkowski, kerr_schild            :258-294          minkowski, kerr_schild  (+ KerrSchild(M, a, textbook))
    dmetric, christoffel, geodesic    :298-370          dmetric, christoffel, geodesic   (evaluated on the GPU)
    Object / Plane / Sphere           :374-428          Plane, Sphere (+ Disk); new subtypes: UserObjects(source)(type, fields)
    Pixel / Canvas / make_canvas      :445-478          Pixel (numpy record), Canvas, make_canvas
    trace_rays(metric, objs, canvas)  :482-536          trace_rays(metric, objs, canvas) -> Canvas
    trace_ray(metric, objs, cb, p)    test/runtests.jl:76   trace_ray(metric, objs, cb, p) -> Pixel
    example1(), example2()            :542-612          example1(), example2()  (write scenes/sphere*.png)

Nothing here computes physics on the CPU: every numeric result comes from librtgr_hip.so (HIP kernels).
"""
import ctypes as C
import math
import os

import numpy as np

from . import _abi
from ._abi import rtgr_camera, rtgr_counters, rtgr_ray_outputs, rtgr_scene, rtgr_solver
from .user_metric import UserMetric, UserObject, UserObjects

D = 4  # src/RayTraceGR.jl:253-254


# ---- metrics (callables only as identities: the ABI takes an enum) ---------------------------------
class Metric:
    """A built-in metric: enum + (M, a). Calling it evaluates g_ab(x) on the GPU (src/RayTraceGR.jl:262-294)."""

    def __init__(self, kind, M=1.0, a=0.0, name="metric", generic=False):
        self.kind, self.M, self.a, self.__name__ = int(kind), float(M), float(a), name
        self.generic = bool(generic)  # True: trace with the generic dual-number RHS (RTGR_METRIC_GENERIC)

    def __call__(self, x, dtype=np.float64):
        g, _, _ = _eval_metric(self, x, want=(True, False, False), dtype=dtype)
        return g

    def __repr__(self):
        return f"{self.__name__}(M={self.M}, a={self.a})"


minkowski = Metric(_abi.MINKOWSKI, name="minkowski")          # src/RayTraceGR.jl:262-264
kerr_schild = Metric(_abi.KS_REF, 1.0, 0.0, name="kerr_schild")  # as written: M=1, a=0 (:275-276), r of :284


def KerrSchild(M=1.0, a=0.0, textbook=True, generic=False):
    """Parameterised Kerr–Schild metric the reference describes (README "varying mass and spin") but does not
    have: textbook radius (RTGR_KS_TRUE) or the as-written radius with a != 0 (RTGR_KS_REF).  generic=True traces with
    the reference-style dual-number RHS instead of the closed contraction (same results, ~5x the flops)."""
    return Metric(_abi.KS_TRUE if textbook else _abi.KS_REF, M, a, name="KerrSchild", generic=generic)


# ---- objects (src/RayTraceGR.jl:374-428) ------------------------------------------------------------------------
class Object:
    kind = 0

    def _pack(self):
        raise NotImplementedError("Called distance on abstract object")  # :384-386


class Plane(Object):
    """Plane{T}(time)  src/RayTraceGR.jl:394-397"""
    kind = _abi.PLANE

    def __init__(self, time):
        self.time = float(time)

    def _pack(self):
        return [self.time] + [0.0] * 8


class Sphere(Object):
    """Sphere{T}(pos, vel, radius)  src/RayTraceGR.jl:409-413 (vel is stored and unused, as in the reference)"""
    kind = _abi.SPHERE

    def __init__(self, pos, vel, radius):
        self.pos = [float(v) for v in pos]
        self.vel = [float(v) for v in vel]
        self.radius = float(radius)
        assert len(self.pos) == D and len(self.vel) == D

    def _pack(self):
        return self.pos + self.vel + [self.radius]


class Disk(Object):
    """Thin disk |z| <= h, r_in <= sqrt(x^2+y^2) <= r_out. No reference counterpart (BASELINE config 5)."""
    kind = _abi.DISK

    def __init__(self, half_thickness, r_in, r_out):
        self.h, self.r_in, self.r_out = float(half_thickness), float(r_in), float(r_out)

    def _pack(self):
        return [self.h, self.r_in, self.r_out] + [0.0] * 6


def make_scene(metric, objs, ctx=None, units=True):
    """(metric, objs::Vector{Object}) -> rtgr_scene (order of objs preserved: it matters, :518-530).  A scene of a
    UserMetric and / or of UserObjects carries the id of its run-time unit in `ctx` (built and loaded on first use), so it
    can only ever run with its own kernels — whichever other units are resident.  units=False leaves the id 0 and touches
    neither compiler nor GPU (a scene description for something else than this library: the tests' CPU oracle)."""
    if not isinstance(metric, (Metric, UserMetric)):
        raise TypeError(
            "a metric is one of the built-ins (minkowski, kerr_schild, KerrSchild(M,a)) or a UserMetric(source) "
            "compiled for the device; a Python callable cannot cross the C ABI (SURVEY §8b)")
    objs = list(objs)
    if len(objs) > _abi.RTGR_OBJECTS_LIMIT:
        raise ValueError(f"at most {_abi.RTGR_OBJECTS_LIMIT} objects")
    # New Object subtypes (src/RayTraceGR.jl:374-389) come as a UserObjects family: its distance / objcolor methods are compiled
    # into ONE unit together with the metric they are traced with (compiled code holds both in the same kernels); the sources of
    # several families are joined into one first (UserObjects.join: type tags renumbered family after family, in the order of
    # first appearance in objs)
    families = list({id(o.family): o.family for o in objs if isinstance(o, UserObject)}.values())
    base = {id(f): 0 for f in families}
    if len(families) > 1:
        joined, bases = UserObjects.join(families)
        base = {id(f): b for f, b in zip(families, bases)}
        families = [joined]
    user_id = 0
    if not units:
        pass
    elif families:
        user_id = families[0].unit_id(metric, ctx)
    elif isinstance(metric, UserMetric):
        user_id = metric.module_id(ctx)
    sc = rtgr_scene()
    sc.metric = metric.kind | (_abi.METRIC_GENERIC if metric.generic else 0)
    sc.nobj, sc.M, sc.a = len(objs), metric.M, metric.a
    sc.user_metric = user_id
    # `objs::Vector{Object{T}}` of any length (:433-441): up to RTGR_MAX_OBJECTS in the scene's inline slots, a longer list as one array
    # behind rtgr_scene.objects (kept alive by the scene: sc._keep)
    slots = sc.obj
    if len(objs) > _abi.RTGR_MAX_OBJECTS:
        slots = sc._keep = (_abi.rtgr_object * len(objs))()
        sc.objects = C.cast(slots, C.POINTER(_abi.rtgr_object))
    for o, obj in enumerate(objs):
        slots[o].kind = obj.kind
        slots[o].type = getattr(obj, "type", 0) + (base[id(obj.family)] if isinstance(obj, UserObject) else 0)
        p = obj._pack()
        for q in range(9):
            slots[o].p[q] = p[q]
    return sc


def eval_objects(metric, objs, x, opt=None, dtype=np.float64, ctx=None):
    """distance(obj, x) of every object, min_distance(objs, x) and the colouring rule of trace_rays at the points x [n, 4], on the
    GPU (rtgr_eval_objects_*; src/RayTraceGR.jl:377-441, :513-533) -> dict(d [n, nobj], dmin [n], hit [n], rgb [n, 3])"""
    lib = _lib()
    sc = make_scene(metric, objs, ctx)
    x = np.ascontiguousarray(x, dtype=dtype).reshape(-1, 4)
    n = x.shape[0]
    opt = opt or solver_defaults(dtype)
    out = dict(d=np.zeros((n, max(sc.nobj, 1)), dtype), dmin=np.zeros(n, dtype), hit=np.zeros(n, np.uint8), rgb=np.zeros((n, 3), dtype))
    fn = lib.rtgr_eval_objects_f64 if dtype == np.float64 else lib.rtgr_eval_objects_f32
    _abi.check(lib, fn(ctx, C.byref(sc), C.byref(opt), x.ctypes.data, n, out["d"].ctypes.data, out["dmin"].ctypes.data,
                       out["hit"].ctypes.data, out["rgb"].ctypes.data))
    out["d"] = out["d"][:, :sc.nobj]
    return out


def check_scene(metric, objs, cam, ni=48, nj=48, opt=None, ctx=None):
    """rtgr_scene_check: the FAR + NEAR passes of THIS scene must deliver the frame of the single FULL pass (every accepted step
    scanned, as the reference does) — the check that catches a rtgr_user_reach that is not an upper bound.  `cam`: make_camera
    arguments (dict) or an rtgr_camera.  Raises RtgrError naming the number of rays that differ; returns None when the frames agree."""
    lib = _lib()
    sc = make_scene(metric, objs, ctx)
    camera = cam if isinstance(cam, rtgr_camera) else make_camera(**cam)
    opt = opt or solver_defaults()
    _abi.check(lib, lib.rtgr_scene_check(ctx, C.byref(sc), C.byref(opt), C.byref(camera), ni, nj, 0))


def solver_defaults(dtype=np.float64, **over):
    """tol = eps(T)^(3/4), λ∈[0,100], hit threshold 0.01, miss colour (1,0,0)  (src/RayTraceGR.jl:485,:497,:519,:528).
    Pure constants — filled here so that building a solver struct does not need the GPU library."""
    s = rtgr_solver()
    tol = float(np.finfo(dtype).eps) ** 0.75
    s.reltol = s.abstol = tol
    s.lambda0, s.lambda1 = 0.0, 100.0
    s.hit_threshold = 0.01
    s.miss_rgb[0], s.miss_rgb[1], s.miss_rgb[2] = 1.0, 0.0, 0.0
    s.max_steps = 100000
    s.interp_points = 10
    for k, v in over.items():
        if k == "miss_rgb":
            for c in range(3):
                s.miss_rgb[c] = float(v[c])
        else:
            setattr(s, k, v)
    return s


def make_camera(pos, widthx, widthy, normal):
    cam = rtgr_camera()
    for a in range(D):
        cam.pos[a], cam.widthx[a], cam.widthy[a], cam.normal[a] = (float(pos[a]), float(widthx[a]),
                                                                   float(widthy[a]), float(normal[a]))
    return cam


# ---- Pixel / Canvas (src/RayTraceGR.jl:445-455) -----------------------------------------------------------------
def pixel_dtype(dtype=np.float64):
    """Pixel{T}: pos (4), normal (4), rgb (3) — isbits, 88 bytes for Float64 (:446-450)."""
    return np.dtype([("pos", dtype, 4), ("normal", dtype, 4), ("rgb", dtype, 3)])


def Pixel(pos, normal, rgb=(0.0, 0.0, 0.0), dtype=np.float64):
    p = np.zeros((), dtype=pixel_dtype(dtype))
    p["pos"], p["normal"], p["rgb"] = pos, normal, rgb
    return p


class Canvas:
    """Canvas{T}(pixels::Array{Pixel{T},2}) — `pixels` is stored column-major like Julia: pixels[i, j] with i fastest
    (numpy array of shape (ni, nj), order='F')."""

    def __init__(self, pixels):
        self.pixels = pixels

    @property
    def shape(self):
        return self.pixels.shape

    def rgb_planes(self):
        """(R, G, B) each (ni, nj) — what `T[p.rgb[c] for p in canvas.pixels]` yields (:566-569)."""
        return tuple(np.asfortranarray(self.pixels["rgb"][..., c]) for c in range(3))

    def image_u8(self):
        """8-bit image[j, i, c] as `save(file, colorview(RGB, R', G', B'))` writes it (:566-575; N0f8 rounding)."""
        rgb = np.stack(self.rgb_planes(), axis=-1)  # [i, j, c]
        img = np.rint(np.clip(rgb, 0.0, 1.0) * 255.0).astype(np.uint8)
        return np.ascontiguousarray(np.transpose(img, (1, 0, 2)))


def _lib():
    lib = _abi.load()
    return lib


def make_canvas(metric, pos, widthx, widthy, normal, ni, nj, dtype=np.float64, ctx=None):
    """make_canvas(metric, pos, widthx, widthy, normal, ni, nj)::Canvas{T}  (src/RayTraceGR.jl:457-478), on the GPU.
    `dtype` plays the reference's type parameter T (np.float64 or np.float32)."""
    lib = _lib()
    sc = make_scene(metric, [], ctx)
    cam = make_camera(pos, widthx, widthy, normal)
    st = np.empty((ni * nj, 8), dtype=dtype)
    fn = lib.rtgr_make_canvas_f64 if dtype == np.float64 else lib.rtgr_make_canvas_f32
    _abi.check(lib, fn(ctx, C.byref(sc), C.byref(cam), ni, nj, 0, nj, st.ctypes.data))
    px = np.zeros((ni, nj), dtype=pixel_dtype(dtype), order="F")
    flat = px.reshape(-1, order="F")
    flat["pos"] = st[:, :4]
    flat["normal"] = st[:, 4:]
    return Canvas(px)


def _canvas_scalar(px_dtype):
    for t in (np.float64, np.float32):
        if px_dtype == pixel_dtype(t):
            return t
    raise TypeError("Canvas{Float64} or Canvas{Float32} expected")


def trace_rays(metric, objs, c, opt=None, return_info=False, ctx=None):
    """trace_rays(metric, objs, c::Canvas{T})::Canvas{T}  (src/RayTraceGR.jl:482-536), T = Float64 or Float32.

    Passes the reference's own AoS pixel array across the ABI (rtgr_trace_pixels_f64 / _f32) and returns a NEW canvas
    with pos/normal copied and rgb set (:532).  Pure, like the reference.  The tolerance is eps(T)^(3/4) (:485).
    `ctx`: an rtgr_context handle (_abi.create_context); on a context of several devices the image rows are dealt
    cyclically to ALL of them inside this one call (the reference's `Threads.@threads` loop, across GPUs)."""
    lib = _lib()
    sc = make_scene(metric, objs, ctx)
    ni, nj = c.pixels.shape
    pin = np.asfortranarray(c.pixels)
    t = _canvas_scalar(pin.dtype)
    opt = opt or solver_defaults(t)
    pout = np.empty_like(pin, order="F")
    ctr = rtgr_counters()
    fn = lib.rtgr_trace_pixels_f64 if t == np.float64 else lib.rtgr_trace_pixels_f32
    _abi.check(lib, fn(ctx, C.byref(sc), C.byref(opt), pin.ctypes.data, ni, nj, pout.ctypes.data, C.byref(ctr)))
    out = Canvas(pout)
    return (out, ctr.as_dict()) if return_info else out


def trace_frames(metric, objs, cams, ni, nj, opt=None, dtype=np.float64, ctx=None, details=False):
    """SEVERAL frames of one scene in one call, two in flight inside the library (rtgr_trace_frames_f64 / _f32) — an extension: the
    reference renders one frame per call (src/RayTraceGR.jl:560, :596).  cams: a list of make_camera argument dicts (or rtgr_camera);
    rays are generated on the device.  -> list of dict(rgb [3, ni*nj], counters (+ the per-ray outputs when details)), frame by frame:
    each the result of the single call (rtgr_trace_f64), bit for bit."""
    lib = _lib()
    sc = make_scene(metric, objs, ctx)
    opt = opt or solver_defaults(dtype)
    K, n = len(cams), ni * nj
    carr = (rtgr_camera * K)(*[c if isinstance(c, rtgr_camera) else make_camera(**c) for c in cams])
    rgb = [np.zeros((3, n), dtype) for _ in range(K)]
    ptrs = (C.c_void_p * K)(*[r.ctypes.data for r in rgb])
    ctrs = (rtgr_counters * K)()
    outs, per = None, [dict() for _ in range(K)]
    if details:
        outs = (rtgr_ray_outputs * K)()
        wide = sc.nobj > 255
        for k in range(K):
            per[k] = dict(state_end=np.zeros((n, 8), dtype), lambda_end=np.zeros(n, dtype), status=np.zeros(n, np.uint8),
                          hit=np.zeros(n, np.uint32 if wide else np.uint8), n_accept=np.zeros(n, np.uint32), n_reject=np.zeros(n, np.uint32))
            for name, arr in per[k].items():
                setattr(outs[k], "hit32" if (wide and name == "hit") else name, arr.ctypes.data)
    fn = lib.rtgr_trace_frames_f64 if dtype == np.float64 else lib.rtgr_trace_frames_f32
    _abi.check(lib, fn(ctx, C.byref(sc), C.byref(opt), K, carr, None, ni, nj, ptrs, outs, ctrs))
    return [dict(per[k], rgb=rgb[k], counters=ctrs[k].as_dict()) for k in range(K)]


def trace_ray(metric, objs, cb, p, opt=None, ctx=None):
    """Legacy single-pixel shape `trace_ray(metric, objs, cb, p)::Pixel` (test/runtests.jl:65-79).
    `cb` is accepted for signature parity and ignored: the callback is always
    ContinuousCallback(min_distance(objs, ·), terminate!) (src/RayTraceGR.jl:488-490)."""
    lib = _lib()
    sc = make_scene(metric, objs, ctx)
    t = _canvas_scalar(p.dtype)
    opt = opt or solver_defaults(t)
    pos = np.ascontiguousarray(p["pos"], dtype=t)
    nrm = np.ascontiguousarray(p["normal"], dtype=t)
    rgb = np.zeros(3, t)
    se = np.zeros(8, t)
    st = C.c_uint8(0)
    fn = lib.rtgr_trace_one_f64 if t == np.float64 else lib.rtgr_trace_one_f32
    _abi.check(lib, fn(ctx, C.byref(sc), C.byref(opt), pos.ctypes.data, nrm.ctypes.data, rgb.ctypes.data, se.ctypes.data,
                       C.addressof(st)))
    return Pixel(pos, nrm, rgb, dtype=t)


# ---- physics kernels for the reference's unit tests (test/runtests.jl:12-61), evaluated on the GPU ------------
def _eval_metric(metric, x, want=(True, True, True), dtype=np.float64):
    lib = _lib()
    sc = make_scene(metric, [])
    x = np.ascontiguousarray(x, dtype=dtype).reshape(-1, 4)
    n = x.shape[0]
    g = np.empty((n, 4, 4), dtype) if want[0] else None
    dg = np.empty((n, 4, 4, 4), dtype) if want[1] else None
    G = np.empty((n, 4, 4, 4), dtype) if want[2] else None
    fn = lib.rtgr_eval_metric_f64 if dtype == np.float64 else lib.rtgr_eval_metric_f32
    _abi.check(lib, fn(None, C.byref(sc), x.ctypes.data, n, g.ctypes.data if want[0] else None,
                       dg.ctypes.data if want[1] else None, G.ctypes.data if want[2] else None))
    sq = (lambda v: v[0] if (v is not None and n == 1) else v)
    return sq(g), sq(dg), sq(G)


def dmetric(metric, x, dtype=np.float64):
    """dmetric(metric, x) -> (g[a,b], dg[a,b,c] = ∂_c g_ab)  (src/RayTraceGR.jl:302-313); dtype = the reference's T"""
    g, dg, _ = _eval_metric(metric, x, (True, True, False), dtype)
    return g, dg


def christoffel(metric, x, dtype=np.float64):
    """christoffel(metric, x) -> Γ[a,b,c] = Γ^a_bc  (src/RayTraceGR.jl:321-331)"""
    return _eval_metric(metric, x, (False, False, True), dtype)[2]


def geodesic(s, metric, lam=0.0, path=0, dtype=np.float64):
    """geodesic(s::SVector{8}, metric, λ) -> ṡ  (src/RayTraceGR.jl:367-370); λ is ignored as in the reference.
    path: 0 closed contraction (IEEE division), 1 generic dual numbers, 2 the integrate loop's own RHS."""
    lib = _lib()
    sc = make_scene(metric, [])
    s = np.ascontiguousarray(s, dtype=dtype).reshape(-1, 8)
    ds = np.empty_like(s)
    fn = lib.rtgr_eval_geodesic_f64 if dtype == np.float64 else lib.rtgr_eval_geodesic_f32
    _abi.check(lib, fn(None, C.byref(sc), s.ctypes.data, s.shape[0], path, ds.ctypes.data))
    return ds[0] if ds.shape[0] == 1 else ds


# ---- example scenes (src/RayTraceGR.jl:542-612) -------------------------------------------------------------------
outdir = "scenes"  # :540


def example1_scene():
    """Scene of example1(): Minkowski, sky R=-10, plane t=-20, sphere R=1/2 at the origin; camera (0,0,-2,0)
    (src/RayTraceGR.jl:545-557).  Returns (metric, objs, camera_args)."""
    caelum = Sphere((0, 0, 0, 0), (1, 0, 0, 0), -10)
    frustum = Plane(-20)
    sphere = Sphere((0, 0, 0, 0), (1, 0, 0, 0), 0.5)
    cam = dict(pos=(0, 0, -2, 0), widthx=(0, 1, 0, 0), widthy=(0, 0, 0, 1), normal=(0, 0, 1, 0))
    return minkowski, [caelum, frustum, sphere], cam


def example2_scene(metric=None):
    """Scene of example2(): kerr_schild, sky R=-10, plane t=-20, sphere R=1/2 at (0,4,0,0); camera (0,4,-2,0)
    (src/RayTraceGR.jl:581-593)."""
    caelum = Sphere((0, 0, 0, 0), (1, 0, 0, 0), -10)
    frustum = Plane(-20)
    sphere = Sphere((0, 4, 0, 0), (1, 0, 0, 0), 0.5)
    cam = dict(pos=(0, 4, -2, 0), widthx=(0, 1, 0, 0), widthy=(0, 0, 0, 1), normal=(0, 0, 1, 0))
    return (metric or kerr_schild), [caelum, frustum, sphere], cam


def _run_example(scene, ni, nj, fname, save, ctx=None):
    from .png import write_png
    metric, objs, cam = scene
    canvas = make_canvas(metric, cam["pos"], cam["widthx"], cam["widthy"], cam["normal"], ni, nj, ctx=ctx)
    canvas = trace_rays(metric, objs, canvas, ctx=ctx)
    if save:
        os.makedirs(outdir, exist_ok=True)
        file = os.path.join(outdir, fname)
        if os.path.exists(file):
            os.remove(file)
        print(f'Output file is "{file}"')
        write_png(file, canvas.image_u8())
    return canvas


def example1(ni=200, nj=200, save=True, ctx=None):
    """example1()  src/RayTraceGR.jl:542-576"""
    return _run_example(example1_scene(), ni, nj, "sphere.png", save, ctx)


def example2(ni=200, nj=200, save=True, ctx=None):
    """example2()  src/RayTraceGR.jl:578-612"""
    return _run_example(example2_scene(), ni, nj, "sphere2.png", save, ctx)


__all__ = ["D", "Metric", "UserMetric", "UserObjects", "UserObject", "minkowski", "kerr_schild", "KerrSchild", "Object", "Plane", "Sphere", "Disk",
           "make_scene", "check_scene", "eval_objects", "solver_defaults", "make_camera", "Pixel", "pixel_dtype", "Canvas", "make_canvas",
           "trace_rays", "trace_ray", "trace_frames", "dmetric", "christoffel", "geodesic", "example1", "example2",
           "example1_scene", "example2_scene"]
