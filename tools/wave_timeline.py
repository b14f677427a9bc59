#!/usr/bin/env python3
"""Per-wave timeline of the persistent FAR pass of ONE launch ("measure the tail"): a -DRTGR_ROOT_STATS build
of the library records {start, end, iterations, rays taken} of every wave (s_memrealtime, 100 MHz); this script builds that
variant, traces one frame (or one rank's cyclic share) and prints

  * how many waves are still running at each tenth of the pass, by age class (0 = oldest wave of its SIMD),
  * per age class: iterations per wave, time per iteration, rays per wave, when the class's last wave ended,
  * lane utilisation = step attempts / (64 x iterations), and what the pass would take at the full-pass throughput with no
    tail (work / throughput of the first half of the pass).

    python tools/wave_timeline.py [--size 1024] [--share 1] [--variant ks_ref0] [--set fair=0,qchunk=8]
"""
import argparse
import ctypes
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--share", type=int, default=1)
    ap.add_argument("--variant", default="ks_ref0")
    ap.add_argument("--set", default="")
    ap.add_argument("--pass", dest="which", default="far", choices=["far", "near"], help="which pass reports its waves")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    vdir = os.path.join(ROOT, "raytracegr.jl_amd", "build", "variants")
    os.makedirs(vdir, exist_ok=True)
    lib_path = os.path.join(vdir, "librtgr_rootstats.so")
    b.build(extra=["-DRTGR_ROOT_STATS"], out=lib_path, obj_dir=os.path.join(vdir, "obj_rootstats"), verbose=False)
    os.environ["RTGR_LIB"] = lib_path
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    rt = load_package()
    from raytracegr_jl_amd import sharded
    import bench
    abi = rt._abi
    lib = abi.load()
    abi.check(lib, lib.rtgr_init(-1))
    lib.rtgr_debug_set_buffer.argtypes = [ctypes.c_void_p]
    sc, cam = bench.build_scene(rt, a.variant)
    opt = rt.solver_defaults()
    j0, st, nr = sharded.row_assignment(a.size, a.share, 0, "cyclic")
    n = a.size * nr
    buf = torch.zeros(4 * 8192 + n + 16, dtype=torch.int64, device="cuda")
    abi.check(lib, lib.rtgr_debug_set_buffer(buf.data_ptr()))
    kw = dict((k, int(v)) for k, v in (kv.split("=") for kv in a.set.split(",") if kv))
    kw["dbg_pass_far"] = 1 if a.which == "far" else 0
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    out = {}
    with abi.options(lib, **kw):
        for _ in range(2):
            sharded.trace_rows_torch(sc, opt, cam, a.size, a.size, j0, st, nr, out=out)
        torch.cuda.synchronize()
        buf.zero_()
        ctr.zero_()
        lib.rtgr_timing_enable(None, 0, 1)
        sharded.trace_rows_torch(sc, opt, cam, a.size, a.size, j0, st, nr, out=out, counters=ctr)
        torch.cuda.synchronize()
        kms, kln = (ctypes.c_double * 4)(), (ctypes.c_uint64 * 4)()
        lib.rtgr_timing_read(None, 0, ctypes.byref(kms), ctypes.byref(kln))
        lib.rtgr_timing_enable(None, 0, 0)
    w = buf[:4 * 8192].cpu().numpy().reshape(8192, 4)
    w = w[w[:, 1] > 0]
    nw = len(w)
    t0 = w[:, 0].min()
    start, end = (w[:, 0] - t0) * 1e-5, (w[:, 1] - t0) * 1e-5   # ms (100 MHz)
    iters, rays = w[:, 2], w[:, 3]
    T = end.max()
    attempts = int(ctr[1] + ctr[2])
    n_simd = 1024
    cls = np.arange(nw) // n_simd
    print(f"{a.variant} {a.size}² share 1/{a.share}: {n / 1e6:.2f} M rays, {attempts / 1e6:.1f} M step attempts, options {kw}")
    print(f"{a.which.upper()} pass timeline: {T:.3f} ms first wave start -> last wave end, {nw} waves ({max(nw // n_simd, 1)} per SIMD); by HIP events: "
          f"FAR {kms[1]:.3f} ms, NEAR {kms[3]:.3f} ms, set-up {kms[0]:.3f}, resolve {kms[2]:.3f}")
    far_attempts = None
    print("waves still running at t/T =  " + "  ".join(f"{f:4.1f}" for f in np.arange(0.1, 1.01, 0.1)))
    for c in range(cls.max() + 1):
        m = cls == c
        print(f"  age class {c} ({m.sum():4d} waves)      " + "  ".join(f"{int((end[m] > f * T).sum()):4d}" for f in np.arange(0.1, 1.01, 0.1)))
    for c in range(cls.max() + 1):
        m = cls == c
        print(f"  class {c}: iterations/wave {iters[m].mean():7.0f} (max {iters[m].max()}), us/iteration {((end[m] - start[m]) / iters[m]).mean() * 1e3:5.2f}, "
              f"rays/wave {rays[m].mean():6.1f}, first wave out {end[m].min():.3f} ms, median {np.median(end[m]):.3f}, last {end[m].max():.3f}")
    lane_steps = 64.0 * iters.sum()
    if a.which == "far":
        print(f"wave-iterations {iters.sum()}, lane utilisation <= {attempts / lane_steps:.3f} (all step attempts of the frame / 64 x FAR iterations; "
              f"the NEAR pass's few are included in the numerator)")
    else:   # per-ray stays of the NEAR pass follow the per-wave block: (accepted steps at hand-over, accepted steps in this pass)
        pr = buf[4 * 8192:4 * 8192 + n].cpu().numpy().view(np.uint32).reshape(-1, 2)[:n]
        stay = pr[:, 1].astype(np.int64)
        print(f"wave-iterations {iters.sum()}; rays that went through the NEAR pass {int((stay > 0).sum())} of {n}, accepted steps taken there "
              f"{int(stay.sum())} ({stay.sum() / max(attempts, 1) * 100:.1f} % of the frame's), mean stay {stay[stay > 0].mean():.1f} steps, "
              f"p99 {np.percentile(stay[stay > 0], 99):.0f}, max {stay.max()}; lane utilisation of the pass {stay.sum() / lane_steps:.3f}")
    # throughput while everybody is busy: iterations per ms in the first half of the pass ~ (sum over waves of iterations done by T/2)
    half = sum(min(1.0, (0.5 * T - s) / max(e - s, 1e-9)) * it for s, e, it in zip(start, end, iters) if s < 0.5 * T)
    rate = half / (0.5 * T)
    print(f"iteration rate in the first half of the pass {rate / 1e3:.1f} k wave-iterations/ms -> the whole pass at that rate: "
          f"{iters.sum() / rate:.3f} ms (tail cost {T - iters.sum() / rate:.3f} ms = {(T - iters.sum() / rate) / T * 100:.1f} %)")


if __name__ == "__main__":
    main()
