"""How well does the queue-order key (256 * sin(angle between the ray and the direction to the origin)) predict the
number of step attempts?  Prints mean / max steps per key bucket for the example2 camera."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import hip_trace
from scenes import rt, scene_variant
import ctypes as C
lib = rt._abi.load(); rt._abi.check(lib, lib.rtgr_init(-1))
for name in sys.argv[1:] or ["ks_ref0"]:
    sc, cam = scene_variant(name)
    n = 512
    s0 = np.zeros((n * n, 8))
    rt._abi.check(lib, lib.rtgr_make_canvas_f64(C.byref(sc), C.byref(cam), n, n, 0, n, s0.ctypes.data))
    out = hip_trace(lib, sc, rt.solver_defaults(), n, n, cam=cam)
    steps = (out["n_accept"] + out["n_reject"]).astype(int)
    x, u = s0[:, 1:4], s0[:, 5:8]
    xx, uu, xu = (x * x).sum(1), (u * u).sum(1), (x * u).sum(1)
    sin2 = np.where(xu < 0, np.maximum(0, 1 - xu * xu / (xx * uu)), 1.0)
    key = np.minimum(255, (256 * np.sqrt(sin2))).astype(int)
    print(name, "steps: mean %.1f max %d" % (steps.mean(), steps.max()))
    print(" bucket  rays   mean   p90   max")
    for b in range(0, 256, 8):
        m = (key >= b) & (key < b + 8)
        if m.any(): print(" %3d-%3d %6d %6.1f %5.0f %5d" % (b, b + 7, m.sum(), steps[m].mean(), np.percentile(steps[m], 90), steps[m].max()))
