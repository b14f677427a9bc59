#!/usr/bin/env python3
"""Float32 (BASELINE config 4) experiment of round 4: the packed two-rays-per-lane kernel as a scan-free FAR pass at
three waves per SIMD + the scalar NEAR pass (option packfar = 1) against the production single FULL pass (packed for a != 0), same
run, interleaved: frame agreement, per-pass kernel times, frame time.      python tools/f32_packfar_ab.py [size] [rounds]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
import torch  # noqa: E402
from raytracegr_jl_amd import sharded  # noqa: E402

abi = rt._abi
lib = abi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
_, objs, cam = rt.example2_scene()
camera = rt.make_camera(**cam)
opt = rt.solver_defaults(np.float32)
for name, metric in (("ks_true08", rt.KerrSchild(1.0, 0.8)), ("ks_ref0", rt.kerr_schild)):
    sc = rt.make_scene(metric, objs)
    frames, times, kms = {}, {0: [], 1: []}, {0: [], 1: []}
    for rnd in range(rounds + 1):
        for pf in (0, 1):
            with abi.options(lib, packfar=pf, pack=1):
                ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
                o = {}
                sharded.trace_rows_torch(sc, opt, camera, n, n, 0, 1, n, dtype=np.float32, counters=ctr, out=o, status=True)   # warm
                torch.cuda.synchronize()
                abi.check(lib, lib.rtgr_timing_enable(None, 0, 1))
                t0 = time.perf_counter()
                for _ in range(10):
                    sharded.trace_rows_torch(sc, opt, camera, n, n, 0, 1, n, dtype=np.float32, counters=ctr, out=o, status=True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10
                ms, ln = (ctypes.c_double * 4)(), (ctypes.c_uint64 * 4)()
                abi.check(lib, lib.rtgr_timing_read(None, 0, ctypes.byref(ms), ctypes.byref(ln)))
                abi.check(lib, lib.rtgr_timing_enable(None, 0, 0))
            if rnd > 0:
                times[pf].append(dt * 1e3)
                kms[pf].append([ms[w] / 10 for w in range(4)])
            frames[pf] = (o["rgb"].cpu().numpy(), o["status"].cpu().numpy(), ctr.cpu().numpy().copy())
    a, b = frames[0], frames[1]
    d = np.abs(a[0] - b[0]).max(axis=0)
    print(f"{name} {n}x{n} f32: frame ms production {np.median(times[0]):.3f} (min {min(times[0]):.3f})  packfar {np.median(times[1]):.3f} (min {min(times[1]):.3f})")
    for pf in (0, 1):
        k = np.median(np.array(kms[pf]), axis=0)
        print(f"   {'packfar ' if pf else 'production'}: setup {k[0]:.3f}  main/FAR {k[1]:.3f}  NEAR {k[3]:.3f}  resolve {k[2]:.3f} ms;  attempts {int(frames[pf][2][1] + frames[pf][2][2]) // 11}  rejected {int(frames[pf][2][2]) // 11}")
    print(f"   frames: status differ {(a[1] != b[1]).sum()}  |drgb| > 1e-3: {(d > 1e-3).sum()} of {d.size}  median {np.median(d):.2e}", flush=True)
