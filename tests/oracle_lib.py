"""Loader + thin numpy wrappers of the CPU oracle (oracle/librtgr_oracle.so). TEST-SIDE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from conftest import ROOT, load_package

rt = load_package()
abi = rt._abi
_ORACLE = os.path.join(ROOT, "oracle", "librtgr_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "rtgr_oracle.cpp")
        hdr = os.path.join(ROOT, "include", "rtgr.h")  # the oracle shares the product's struct layouts
        if (not os.path.exists(_ORACLE)) or os.path.getmtime(_ORACLE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        # (RTGR_ORACLE_LIB: another build of the same source — tools/sanitize_host.sh runs the CPU tests against an ASan / UBSan one)
        _lib = C.CDLL(os.environ.get("RTGR_ORACLE_LIB") or _ORACLE)
    return _lib


def _outs(n, dtype, want, wide=False):
    """wide: the hit map as 32 bits (rtgr_ray_outputs.hit32) — object lists beyond 255"""
    o = abi.rtgr_ray_outputs()
    arrs = {}
    if want:
        arrs = dict(state_end=np.zeros((n, 8), dtype), lambda_end=np.zeros(n, dtype), status=np.zeros(n, np.uint8),
                    hit=np.zeros(n, np.uint32 if wide else np.uint8), n_accept=np.zeros(n, np.uint32), n_reject=np.zeros(n, np.uint32))
        for k, v in arrs.items():
            setattr(o, "hit32" if (wide and k == "hit") else k, v.ctypes.data)
    return o, arrs


def trace(scene, opt, ni, nj, j0=0, j1=None, cam=None, state0=None, dtype=np.float64, nthreads=0, details=True):
    """Oracle twin of rtgr_trace_f64/f32. Returns dict(rgb[3,n], counters, + per-ray outputs)."""
    j1 = nj if j1 is None else j1
    n = ni * (j1 - j0)
    rgb = np.zeros((3, n), dtype)
    o, arrs = _outs(n, dtype, details, wide=scene.nobj > 255)
    ctr = abi.rtgr_counters()
    fn = lib().rtgr_oracle_trace_f64 if dtype == np.float64 else lib().rtgr_oracle_trace_f32
    s0 = None
    if state0 is not None:
        state0 = np.ascontiguousarray(state0, dtype)
        s0 = C.c_void_p(state0.ctypes.data)
    rc = fn(C.byref(scene), C.byref(opt), s0, C.byref(cam) if cam is not None else None, C.c_uint64(ni),
            C.c_uint64(nj), C.c_uint64(j0), C.c_uint64(j1), C.c_void_p(rgb.ctypes.data), C.byref(o), C.byref(ctr),
            C.c_int(nthreads))
    assert rc == 0, rc
    arrs.update(rgb=rgb, counters=ctr.as_dict())
    return arrs


def make_canvas(scene, cam, ni, nj, j0=0, j1=None, dtype=np.float64):
    j1 = nj if j1 is None else j1
    st = np.zeros((ni * (j1 - j0), 8), dtype)
    fn = lib().rtgr_oracle_make_canvas_f64 if dtype == np.float64 else lib().rtgr_oracle_make_canvas_f32
    rc = fn(C.byref(scene), C.byref(cam), C.c_uint64(ni), C.c_uint64(nj), C.c_uint64(j0), C.c_uint64(j1),
            C.c_void_p(st.ctypes.data))
    assert rc == 0
    return st


def eval_metric(scene, x, dtype=np.float64):
    x = np.ascontiguousarray(x, dtype).reshape(-1, 4)
    n = x.shape[0]
    g, dg, G = np.zeros((n, 4, 4), dtype), np.zeros((n, 4, 4, 4), dtype), np.zeros((n, 4, 4, 4), dtype)
    fn = lib().rtgr_oracle_eval_metric_f64 if dtype == np.float64 else lib().rtgr_oracle_eval_metric_f32
    rc = fn(C.byref(scene), C.c_void_p(x.ctypes.data), C.c_uint64(n), C.c_void_p(g.ctypes.data),
            C.c_void_p(dg.ctypes.data), C.c_void_p(G.ctypes.data))
    assert rc == 0
    return g, dg, G


def metric_plain(scene, x):
    x = np.ascontiguousarray(x, np.float64).reshape(-1, 4)
    g = np.zeros((x.shape[0], 4, 4))
    assert lib().rtgr_oracle_metric_plain_f64(C.byref(scene), C.c_void_p(x.ctypes.data), C.c_uint64(x.shape[0]),
                                              C.c_void_p(g.ctypes.data)) == 0
    return g


def inv4(m):
    m = np.ascontiguousarray(m, np.float64)
    o = np.zeros((4, 4))
    lib().rtgr_oracle_inv4_f64(C.c_void_p(m.ctypes.data), C.c_void_p(o.ctypes.data))
    return o


def geodesic(scene, s, long_double=False):
    s = np.ascontiguousarray(s, np.float64).reshape(-1, 8)
    ds = np.zeros_like(s)
    fn = lib().rtgr_oracle_eval_geodesic_ld if long_double else lib().rtgr_oracle_eval_geodesic_f64
    assert fn(C.byref(scene), C.c_void_p(s.ctypes.data), C.c_uint64(s.shape[0]), C.c_void_p(ds.ctypes.data)) == 0
    return ds


def image_u8(rgb, ni, nj):
    """rgb[3, ni*nj] (linear index i + j*ni) -> uint8 image[j, i, c] with N0f8 rounding."""
    a = np.rint(np.clip(rgb, 0, 1) * 255.0).astype(np.uint8).reshape(3, nj, ni)
    return np.ascontiguousarray(np.transpose(a, (1, 2, 0)))


def eval_objects(scene, opt, x, dtype=np.float64):
    """oracle twin of rtgr_eval_objects_*: dict(d [n, nobj], dmin [n], hit [n], rgb [n, 3])"""
    x = np.ascontiguousarray(x, dtype).reshape(-1, 4)
    n = x.shape[0]
    out = dict(d=np.zeros((n, max(scene.nobj, 1)), dtype), dmin=np.zeros(n, dtype), hit=np.zeros(n, np.uint8), rgb=np.zeros((n, 3), dtype))
    fn = lib().rtgr_oracle_eval_objects_f64 if dtype == np.float64 else lib().rtgr_oracle_eval_objects_f32
    rc = fn(C.byref(scene), C.byref(opt), C.c_void_p(x.ctypes.data), C.c_uint64(n), C.c_void_p(out["d"].ctypes.data),
            C.c_void_p(out["dmin"].ctypes.data), C.c_void_p(out["hit"].ctypes.data), C.c_void_p(out["rgb"].ctypes.data))
    assert rc == 0
    out["d"] = out["d"][:, :scene.nobj]
    return out
