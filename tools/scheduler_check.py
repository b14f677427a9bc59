#!/usr/bin/env python3
"""Are the scheduler's size-keyed choices (longest-first order key, FAR4 threshold, priority rotation, early list) fitted
to example2's camera?  (VERDICT r1 #13.)  Times the pipeline for OTHER cameras of the same scene with each choice forced
on / off next to the automatic setting.  The choices never change a result bit (tests), only the time.

    python tools/scheduler_check.py [size ...]          # default sizes 1024 4096
Cameras: example2 (the tuning camera: (4,-2,0), looking along +y past the hole); far20 (r = 20, off-axis, looking at the
hole: nearly every ray is a short sky ray, the hole is a small disc in the middle); inside (r = 2.6, looking tangentially:
a third of the rays start next to the photon region and are long).
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
from raytracegr_jl_amd import sharded  # noqa: E402

abi = rt._abi
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))


def cam_lookat(pos, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), width=1.0):
    p, t, upv = np.array(pos, float), np.array(target, float), np.array(up, float)
    n = (t - p) / np.linalg.norm(t - p)
    x = np.cross(n, upv)
    x /= np.linalg.norm(x)
    y = np.cross(x, n)
    return dict(pos=(0, *p), widthx=(0, *(width * x)), widthy=(0, *(width * y)), normal=(0, *n))


CAMS = {"example2": rt.example2_scene()[2], "far20": cam_lookat((14.0, -14.0, 3.0)),
        "inside": cam_lookat((2.6, 0.0, 0.0), target=(2.6, 5.0, 0.0))}
SETTINGS = [("auto", {}), ("order=0", {"order": 0}), ("fair=0", {"fair": 0}), ("fair=13", {"fair": 13}), ("far4=0", {"far4": 0}),
            ("far4=1", {"far4": 1}), ("near_early=0", {"near_early": 0}), ("waves_per_cu_near=8", {"waves_per_cu_near": 8}),
            ("waves_per_cu_near=4", {"waves_per_cu_near": 4}), ("qchunk=64", {"qchunk": 64}), ("qchunk=256", {"qchunk": 256})]


def run(sc, opt, cam, n, reps=3):
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    out = {}
    sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, counters=ctr, out=out)
    torch.cuda.synchronize()
    ctr.zero_()
    t0 = time.perf_counter()
    for _ in range(reps):
        sharded.trace_rows_torch(sc, opt, cam, n, n, 0, 1, n, counters=ctr, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e3, (int(ctr[1]) + int(ctr[2])) / reps / (n * n)


sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096]
for variant in ("ks_ref0", "ks_true08"):
    metric = rt.kerr_schild if variant == "ks_ref0" else rt.KerrSchild(1, 0.8)
    _, objs, _ = rt.example2_scene()
    sc, opt = rt.make_scene(metric, objs), rt.solver_defaults()
    for cname, cam in CAMS.items():
        camera = rt.make_camera(**cam)
        for n in sizes:
            row = []
            for sname, kn in SETTINGS:
                if "far4" in kn and variant != "ks_ref0":
                    continue
                with abi.options(lib, **kn):
                    ms, spr = run(sc, opt, camera, n)
                row.append(f"{sname} {ms:.2f}")
            print(f"{variant:9s} {cname:9s} {n:5d}²  {spr:6.1f} steps/ray | " + " | ".join(row), flush=True)
