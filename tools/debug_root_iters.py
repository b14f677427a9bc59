"""Histogram of root-find iterations per ray (needs a -DRTGR_ROOT_STATS build: RTGR_LIB=raytracegr.jl_amd/build/librtgr_hip_stats.so)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import hip_trace
from scenes import rt, scene_variant
lib = rt._abi.load(); rt._abi.check(lib, lib.rtgr_init(-1))
for name in sys.argv[1:] or ["ks_ref0"]:
    sc, cam = scene_variant(name)
    out = hip_trace(lib, sc, rt.solver_defaults(), 512, 512, cam=cam)
    it = out["lambda_end"].astype(int)
    print(name, "mean", it.mean(), "max", it.max(), "hist", np.bincount(it)[:40])
    w = it.reshape(-1, 64).max(axis=1)
    print("   per-wave max: mean", w.mean(), "hist", np.bincount(w)[:40])
    for h in (1, 2, 3):
        m = out["hit"] == h
        print("   hit", h, "n", m.sum(), "mean iters", it[m].mean() if m.any() else 0, "max", it[m].max() if m.any() else 0)
    slow = it > 12
    print("   slow rays:", slow.sum(), "hit classes", np.bincount(out["hit"][slow], minlength=4), "status", np.bincount(out["status"][slow]))
    ii = np.nonzero(slow)[0][:12]
    for q in ii:
        print("     ray", q % 512, q // 512, "iters", it[q], "hit", out["hit"][q], "x_end", out["state_end"][q, :4], "nacc", out["n_accept"][q])
