#!/usr/bin/env python3
"""Three calls of rtgr_trace_pixels_f64 at 4096² (for a rocprofv3 --kernel-trace --memory-copy-trace timeline)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
abi = rt._abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
metric, objs, camd = rt.example2_scene()
sc, cam, opt = rt.make_scene(metric, objs), rt.make_camera(**camd), rt.solver_defaults()
st = np.empty((n * n, 8))
abi.check(lib, lib.rtgr_make_canvas_f64(None, C.byref(sc), C.byref(cam), n, n, 0, n, st.ctypes.data))
px = np.zeros(n * n, dtype=rt.pixel_dtype())
px["pos"], px["normal"] = st[:, :4], st[:, 4:]
out = np.empty_like(px)
for k in range(3):
    t0 = time.perf_counter()
    abi.check(lib, lib.rtgr_trace_pixels_f64(None, C.byref(sc), C.byref(opt), px.ctypes.data, n, n, out.ctypes.data, None))
    print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
