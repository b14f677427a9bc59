"""Predict the strong-scaling curve on one GPU: time the rank-0 slab of a 4096² frame split over N = 1,2,4,8 ranks."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rt = load_package()
from raytracegr_jl_amd import sharded
metric, objs, cam = rt.example2_scene()
sc, opt, camera = rt.make_scene(metric, objs), rt.solver_defaults(), rt.make_camera(**cam)
ni = nj = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t1 = None
for N in (1, 2, 4, 8):
    ts = []
    layout = sys.argv[2] if len(sys.argv) > 2 else "cyclic"
    for r in sorted({0, N // 2, N - 1}):
        j0, st, nr = sharded.row_assignment(nj, N, r, layout)
        out = {}
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sharded.trace_rows_torch(sc, opt, camera, ni, nj, j0, st, nr, out=out)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ts.append(dt * 1e3)
    import ctypes
    lib = rt._abi.load()
    lib.rtgr_timing_enable(None, 0, 1)
    sharded.trace_rows_torch(sc, opt, camera, ni, nj, j0, st, nr, out=out)
    torch.cuda.synchronize()
    kms = (ctypes.c_double * 4)(); kln = (ctypes.c_uint64 * 4)()
    lib.rtgr_timing_read(None, 0, ctypes.byref(kms), ctypes.byref(kln))
    lib.rtgr_timing_enable(None, 0, 0)
    print("      kernels ms: canvas+order %.2f  FAR %.2f  NEAR %.2f  resolve %.2f" % (kms[0], kms[1], kms[3], kms[2]))
    j1 = j0 + nr
    t = max(ts)
    t1 = t1 or t
    print(f"N={N} {layout}: {ni}x{nr} rays {ni*nr/1e6:.2f}M  time per rank (max of ranks 0/mid/last) {t:.2f} ms  speedup vs N=1 {t1/t:.2f}x  eff {t1/t/N:.2f}  {[round(x,2) for x in ts]}")
