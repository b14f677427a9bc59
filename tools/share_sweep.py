#!/usr/bin/env python3
"""Launch-option sweep on ONE rank's share of a 4096² frame split N ways (cyclic rows): FAR / NEAR / total ms per option
set.  What the N-GPU bench runs per rank, rehearsed on one GPU.

    python tools/share_sweep.py [N=8] [variant=ks_ref0]
"""
import ctypes
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from raytracegr_jl_amd import sharded  # noqa: E402

abi = rt._abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = sys.argv[2] if len(sys.argv) > 2 else "ks_ref0"
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sc, cam = bench.build_scene(rt, variant)
opt = rt.solver_defaults()
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
ni = nj = 4096
j0, st, nr = sharded.row_assignment(nj, N, 0, "cyclic")
out = {}


def run(reps=5, **kw):
    with abi.options(lib, **kw):
        for _ in range(2):
            sharded.trace_rows_torch(sc, opt, cam, ni, nj, j0, st, nr, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            sharded.trace_rows_torch(sc, opt, cam, ni, nj, j0, st, nr, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        lib.rtgr_timing_enable(None, 0, 1)
        sharded.trace_rows_torch(sc, opt, cam, ni, nj, j0, st, nr, out=out)
        torch.cuda.synchronize()
        kms = (ctypes.c_double * 4)()
        kln = (ctypes.c_uint64 * 4)()
        lib.rtgr_timing_read(None, 0, ctypes.byref(kms), ctypes.byref(kln))
        lib.rtgr_timing_enable(None, 0, 0)
    return dt, kms[0], kms[1], kms[3], kms[2]


print(f"N={N} share of 4096² ({ni * nr / 1e6:.2f} M rays), {variant}")
print(f"{'options':44s} total   setup  FAR    NEAR   resolve")
sets = [{}]
for k, vals in (("far4", (0, 1)), ("waves_per_cu", (8, 12, 16)), ("fair", (0, 12, 13, 14)), ("qchunk", (16, 32, 64, 128, 256)),
                ("waves_per_cu_near", (4, 8, 12)), ("qchunk_near", (32, 64, 128, 256)), ("near_early", (0, 16, 32, 128, 256)),
                ("order", (0,)), ("split", (0,))):
    for v in vals:
        sets.append({k: v})
sets += [{"far4": 1, "waves_per_cu": 16, "fair": 13}, {"waves_per_cu": 8, "waves_per_cu_near": 8}]
for kw in sets:
    try:
        r = run(**kw)
    except Exception as e:  # noqa: BLE001
        print(f"{str(kw):44s} ERROR {e}")
        continue
    print(f"{str(kw):44s} {r[0]:6.2f}  {r[1]:5.2f}  {r[2]:6.2f} {r[3]:5.2f}  {r[4]:5.2f}", flush=True)
