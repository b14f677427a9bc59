#!/usr/bin/env python3
"""Global integration error of the HIP path against the true geodesics of tests/golden/truth_*.npz (made by
tests/golden/make_truth.py; no oracle involved): per scene variant, the largest |end state|, |λ_end| and wrap-aware RGB
difference over the sampled pixels, for the closed-form kernels, the generic dual-number RHS and the Float32 path.

    python tools/global_error.py            # needs the GPU
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenes import rt, scene_variant  # noqa: E402  (scene builders only)

abi = rt._abi
VARIANTS = ["mink", "ks_ref0", "ks_ref08", "ks_true0", "ks_true08", "ks_true0998", "ks_true0998_disk"]


def trace(lib, sc, cam, n, dtype):
    nr = n * n
    rgb = np.zeros((3, nr), dtype)
    out = dict(state_end=np.zeros((nr, 8), dtype), lambda_end=np.zeros(nr, dtype), hit=np.zeros(nr, np.uint8))
    o = abi.rtgr_ray_outputs()
    for k, v in out.items():
        setattr(o, k, v.ctypes.data)
    opt = rt.solver_defaults(dtype)
    fn = lib.rtgr_trace_f64 if dtype == np.float64 else lib.rtgr_trace_f32
    abi.check(lib, fn(None, C.byref(sc), C.byref(opt), None, C.byref(cam), n, n, 0, n, rgb.ctypes.data, C.byref(o), None))
    out["rgb"] = rgb
    return out


def errors(got, f):
    n = int(f["n"])
    p = f["ij"][:, 0] + n * f["ij"][:, 1]
    same = got["hit"][p] == f["hit"]
    ds = np.abs(got["state_end"][p] - f["state_end"]).max(axis=1)
    dl = np.abs(got["lambda_end"][p] - f["lambda_end"])
    d = np.abs(got["rgb"][:, p].T - f["rgb"])
    per = (f["hit"] / 3.0)[:, None]
    d = np.minimum(d, np.abs(per - d)).max(axis=1)
    sph, cap = same & (f["hit"] != 2), same & (f["hit"] == 2)
    mx = lambda a, m: float(a[m].max()) if m.any() else 0.0
    return int((~same).sum()), mx(ds, sph), mx(dl, sph), mx(d, sph), mx(ds, cap), mx(dl, cap)


if __name__ == "__main__":
    lib = abi.load()
    abi.check(lib, lib.rtgr_init(-1))
    print(f"{'variant':18s} {'path':8s} flips  sphere-hit rays: |Δstate| |Δλ| |ΔRGB|     captured rays: |Δstate| |Δλ|")
    for name in VARIANTS:
        f = np.load(os.path.join(ROOT, "tests", "golden", f"truth_{name}.npz"))
        n = int(f["n"])
        for path, dtype, flag in (("closed", np.float64, 0), ("generic", np.float64, abi.METRIC_GENERIC),
                                  ("f32", np.float32, 0)):
            sc, cam = scene_variant(name)
            if flag and name == "mink":
                continue
            sc.metric |= flag
            e = errors(trace(lib, sc, cam, n, dtype), f)
            print(f"{name:18s} {path:8s} {e[0]:4d}   {e[1]:.1e} {e[2]:.1e} {e[3]:.1e}      {e[4]:.1e} {e[5]:.1e}", flush=True)
