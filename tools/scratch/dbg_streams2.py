import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenes import rt, scene_variant, example
from raytracegr_jl_amd import sharded
import oracle_lib as O
abi = rt._abi
lib = abi.load(); abi.check(lib, lib.rtgr_init(-1))
def hip_trace(sc, opt, ni, nj, cam):
    n = ni * nj
    rgb = np.zeros((3, n)); o, arrs = O._outs(n, np.float64, True); ctr = abi.rtgr_counters()
    abi.check(lib, lib.rtgr_trace_f64(None, C.byref(sc), C.byref(opt), None, C.byref(cam), ni, nj, 0, nj, rgb.ctypes.data, C.byref(o), C.byref(ctr)))
    return rgb
if "host" in sys.argv:
    for name in ("ks_ref0", "ks_true08", "mink"):
        sc, cam = scene_variant(name)
        hip_trace(sc, rt.solver_defaults(), 32, 32, cam)
def tr(sc, opt, cam, ni, nj, stream=None):
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
        out = sharded.trace_slab_torch(sc, opt, cam, ni, nj, 0, nj, details=True, counters=ctr)
    out["ctr"] = ctr
    return out
jobs = [(scene_variant("ks_ref0"), 320, 256), (scene_variant("ks_true0998_disk"), 256, 320)]
opt = rt.solver_defaults()
serial = []
for (sc, cam), ni, nj in jobs:
    serial.append(tr(sc, opt, cam, ni, nj)); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("streams", hex(s1.cuda_stream), hex(s2.cuda_stream), hex(torch.cuda.current_stream().cuda_stream))
for rep in range(3):
    outs = []
    for ((sc, cam), ni, nj), st in zip(jobs, (s1, s2)):
        outs.append(tr(sc, opt, cam, ni, nj, stream=st))
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(serial, outs)):
        for key in a:
            x, y = a[key], b[key]
            bad = ~((x == y) | (x.isnan() & y.isnan())) if x.is_floating_point() else (x != y)
            if bad.any():
                idx = bad.nonzero()
                print("rep", rep, "job", k, key, "n_bad", int(bad.sum()), "of", bad.numel(), "first", idx[0].tolist(), "last", idx[-1].tolist())
                if key == "rgb":
                    i = int(idx[0][-1])
                    for kk in ("status", "hit", "n_accept", "n_reject", "lambda_end"):
                        print("   ", kk, a[kk][i].item(), b[kk][i].item())
                    print("    rgb", a["rgb"][:, i].tolist(), b["rgb"][:, i].tolist())
print("done")
