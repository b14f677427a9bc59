"""Dump the event record of the slowest-converging rays and replay the root finder in numpy
(needs RTGR_LIB=raytracegr.jl_amd/build/librtgr_hip_stats.so, a -DRTGR_ROOT_STATS build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import hip_trace
from scenes import rt, scene_variant
lib = rt._abi.load(); rt._abi.check(lib, lib.rtgr_init(-1))
sc, cam = scene_variant("ks_ref0")
n = 256
out = hip_trace(lib, sc, rt.solver_defaults(), n, n, cam=cam)      # details=True -> recw = 44
it = out["lambda_end"].astype(int)
rec = np.zeros((n * n, 44))
lib.rtgr_debug_workspace.argtypes = [C.c_void_p, C.c_uint64]
assert lib.rtgr_debug_workspace(rec.ctypes.data, rec.nbytes) == 0
eps = 2.220446049250313e-16
objs = [(2, 0, 0, 0, -10.0), (1, -20.0), (2, 4.0, 0, 0, 0.5)]
def cond(x):
    d = []
    for o in objs:
        if o[0] == 1: d.append(x[0] - o[1])
        else:
            dd = (x[1]-o[1])**2 + (x[2]-o[2])**2 + (x[3]-o[3])**2 - o[4]**2
            d.append(-dd if o[4] < 0 else dd)
    return min(d), d
for q in np.argsort(-it)[:4]:
    r = rec[q]; x0 = r[0:4]; c = r[4:20].reshape(4, 4); ps, top, t, h = r[20:24]
    pos = lambda th: x0 + th * (c[0] + th * (c[1] + th * (c[2] + th * c[3])))
    print("ray", q, "iters", it[q], "ps", ps, "top", top, "h", h, "hit", out["hit"][q])
    print("   cond(0)", cond(pos(0.0)), "cond(top)", cond(pos(top)))
    ths = np.linspace(0, top, 9)
    print("   cond on grid", [float("%.3g" % cond(pos(th))[0]) for th in ths])
    # locate the root by bisection in numpy and print the neighbourhood
    lo, hi = 0.0, top
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if cond(pos(mid))[0] * ps > 0: lo = mid
        else: hi = mid
    print("   root ~", lo, "width", hi - lo, " cond just around:", [float("%.3g" % cond(pos(lo + k * 4 * eps * max(lo, top / 16)))[0]) for k in range(-6, 7)])
sel = np.argsort(-it)[:8]
np.savez(os.path.join(ROOT, "gpurun_out", "slow_rays.npz"), rec=rec[sel], iters=it[sel], lam=out["lambda_end"][sel])
