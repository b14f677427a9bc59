#!/usr/bin/env python3
"""Wall time of the single-process multi-device entry points at 4096² on a context that lists this GPU N times (logical
devices: streams, workspaces and copies of their own): rtgr_trace_sharded_device_f64 (frame left on device 0) against
rtgr_trace_sharded_f64 (frame downloaded to host memory), and the plain single-device host entry for reference.

    python tools/sharded_host_timing.py [N=2] [size=4096]
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
abi = rt._abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
metric, objs, camd = rt.example2_scene()
sc, cam, opt = rt.make_scene(metric, objs), rt.make_camera(**camd), rt.solver_defaults()
ctx = abi.create_context(lib, [torch.cuda.current_device()] * N)
rgb = np.empty((3, n * n))
d_rgb = torch.empty((3, n * n), dtype=torch.float64, device="cuda")
ctr = abi.rtgr_counters()


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


dev = timed(lambda: abi.check(lib, lib.rtgr_trace_sharded_device_f64(ctx, C.byref(sc), C.byref(opt), C.byref(cam), n, n, d_rgb.data_ptr(), None, C.byref(ctr))))
host = timed(lambda: abi.check(lib, lib.rtgr_trace_sharded_f64(ctx, C.byref(sc), C.byref(opt), C.byref(cam), n, n, rgb.ctypes.data, None, C.byref(ctr))))
one = timed(lambda: abi.check(lib, lib.rtgr_trace_f64(None, C.byref(sc), C.byref(opt), None, C.byref(cam), n, n, 0, n, rgb.ctypes.data, None, C.byref(ctr))))
print(f"{N} logical devices, {n}x{n}: sharded_device {dev:.1f} ms | sharded (host frame) {host:.1f} ms | single-device host entry {one:.1f} ms")
abi.check(lib, lib.rtgr_destroy(ctx))
