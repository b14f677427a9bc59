"""Very long lists (10^3 … 10^5 small spheres in a cloud around the hole, 256² canvas): where a frame's time goes.   python tools/big_lists.py"""
import sys, time, ctypes as C
sys.path.insert(0, "tests")
import numpy as np
import conftest; conftest.load_package()
from raytracegr_jl_amd import api as rt, _abi as abi
lib = abi.load(); abi.check(lib, lib.rtgr_init(-1))
metric, objs3, cam = rt.example2_scene()
camera = rt.make_camera(**cam); opt = rt.solver_defaults()
for n in (1000, 10000, 100000):
    rng = np.random.default_rng(1)
    c = rng.normal(size=(n, 3)) * 4.0
    bad = np.linalg.norm(c, axis=1) < 2.8
    c[bad] *= (3.0 / np.linalg.norm(c[bad], axis=1))[:, None]
    sc0 = rt.make_scene(rt.kerr_schild, objs3[:2])
    arr = (abi.rtgr_object * (n + 2))(); arr[0], arr[1] = sc0.obj[0], sc0.obj[1]
    view = np.frombuffer(arr, dtype=np.float64).reshape(n + 2, C.sizeof(abi.rtgr_object) // 8)
    kinds = np.frombuffer(arr, dtype=np.uint32).reshape(n + 2, C.sizeof(abi.rtgr_object) // 4)
    p0 = abi.rtgr_object.p.offset // 8
    kinds[2:, abi.rtgr_object.kind.offset // 4] = abi.SPHERE
    view[2:, p0 + 1:p0 + 4] = c; view[2:, p0 + 8] = 0.3 * n ** (-1 / 3)
    sc = sc0.clone(); sc.objects, sc.nobj = C.cast(arr, C.POINTER(abi.rtgr_object)), n + 2
    for size in (256,):
        rgb = np.zeros((3, size * size)); hit = np.zeros(size * size, np.uint32); o = abi.rtgr_ray_outputs(); o.hit32 = hit.ctypes.data
        ctr = abi.rtgr_counters()
        for rep in range(2):
            if rep == 1:
                abi.check(lib, lib.rtgr_timing_enable(None, 0, 1))
            t = time.time()
            abi.check(lib, lib.rtgr_trace_f64(None, C.byref(sc), C.byref(opt), None, C.byref(camera), size, size, 0, size, rgb.ctypes.data, C.byref(o), C.byref(ctr)))
            dt = time.time() - t
        kms, kln = (C.c_double * 4)(), (C.c_uint64 * 4)()
        abi.check(lib, lib.rtgr_timing_read(None, 0, C.byref(kms), C.byref(kln)))
        abi.check(lib, lib.rtgr_timing_enable(None, 0, 0))
        print(f"{n} spheres, {size}²: {dt*1e3:.1f} ms per call (second call): set-up {kms[0]:.1f}  far {kms[1]:.1f}  near {kms[3]:.1f}  resolve {kms[2]:.1f} ms; "
              f"{(ctr.accepted + ctr.rejected) / size / size:.1f} steps/ray, {len(np.unique(hit))} objects on screen", flush=True)
