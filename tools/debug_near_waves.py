"""Per-wave timeline of the NEAR pass (RTGR_DBG_PASS=far: of the FAR pass) (needs RTGR_LIB=raytracegr.jl_amd/build/librtgr_hip_stats.so, -DRTGR_ROOT_STATS)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rt = load_package()
from raytracegr_jl_amd import sharded
lib = rt._abi.load(); rt._abi.check(lib, lib.rtgr_init(-1))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenes import scene_variant
sc, camera = scene_variant(os.environ.get("RTGR_VARIANT", "ks_ref0"))   # BASELINE scene variants, tests/scenes.py
DT = np.float32 if os.environ.get("RTGR_DTYPE") == "f32" else np.float64
opt = rt.solver_defaults(DT)
NMAX = 4096 * 4096
buf = torch.zeros(4 * 8192 + NMAX, dtype=torch.int64, device="cuda")   # per-wave records, then 2 x u32 per ray
lib.rtgr_debug_set_buffer.argtypes = [C.c_void_p]
lib.rtgr_debug_set_buffer(buf.data_ptr())
for n in [int(a) for a in sys.argv[1:]] or [4096]:
    for rep in range(2):
        buf.zero_()
        sharded.trace_slab_torch(sc, opt, camera, n, n, 0, n, dtype=DT)
        torch.cuda.synchronize()
    d = buf[:4 * 8192].cpu().numpy().reshape(-1, 4)
    d = d[d[:, 1] > 0]
    t0 = d[:, 0].min()
    st, en = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0     # wall_clock64: 100 MHz -> µs
    print(f"{n}x{n}: {len(d)} waves; kernel span {en.max():.0f} us; wave start: median {np.median(st):.0f} max {st.max():.0f} us; "
          f"wave end: p10 {np.percentile(en,10):.0f} median {np.median(en):.0f} p90 {np.percentile(en,90):.0f} p99 {np.percentile(en,99):.0f} max {en.max():.0f} us")
    print(f"   alive fraction {((en-st).sum()/len(d))/en.max():.2f}; iterations per wave: median {np.median(d[:,2]):.0f} max {d[:,2].max()}; rays per wave: median {np.median(d[:,3]):.0f} min {d[:,3].min()} max {d[:,3].max()}")
    print(f"   us per iteration (median wave): {np.median((en-st)/np.maximum(d[:,2],1)):.2f}")
    upi = (en - st) / np.maximum(d[:, 2], 1)
    nw = len(d)
    print("   us/iteration by wave-id octile:", " ".join(f"{upi[k*nw//8:(k+1)*nw//8].mean():.2f}" for k in range(8)))
    print("   iterations by wave-id octile:  ", " ".join(f"{d[k*nw//8:(k+1)*nw//8, 2].mean():.0f}" for k in range(8)))
    print("   us/iteration percentiles: p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f" % tuple(np.percentile(upi, [1, 10, 50, 90, 99])))
    late = np.argsort(-en)[:5]
    for w in late: print(f"   late wave {w}: start {st[w]:.0f} end {en[w]:.0f} iters {d[w,2]} rays {d[w,3]}")
    if os.environ.get("RTGR_DBG_PASS") == "far" or DT == np.float32:
        continue
    pr = buf[4 * 8192:4 * 8192 + n * n].cpu().numpy().view(np.uint32).reshape(-1, 2)
    n0, stay = pr[:, 0].astype(int), pr[:, 1].astype(int)
    print(f"   NEAR stay: mean {stay.mean():.2f} p50 {np.percentile(stay,50):.0f} p90 {np.percentile(stay,90):.0f} p99 {np.percentile(stay,99):.0f} p99.9 {np.percentile(stay,99.9):.0f} max {stay.max()}")
    for thr in (20, 50, 100):
        m = stay >= thr
        print(f"   rays staying >= {thr}: {m.sum()} ({100*m.mean():.2f} %), their steps at hand-over: min {n0[m].min() if m.any() else 0} p50 {np.percentile(n0[m],50) if m.any() else 0:.0f} p90 {np.percentile(n0[m],90) if m.any() else 0:.0f} max {n0[m].max() if m.any() else 0}; share of NEAR steps {100*stay[m].sum()/stay.sum():.1f} %")
    print(f"   steps at hand-over overall: p1 {np.percentile(n0,1):.0f} p10 {np.percentile(n0,10):.0f} p50 {np.percentile(n0,50):.0f}")
