import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenes import rt, scene_variant
from raytracegr_jl_amd import sharded
abi = rt._abi
lib = abi.load(); abi.check(lib, lib.rtgr_init(-1))
def tr(sc, opt, cam, ni, nj, stream=None, details=True):
    with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
        ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
        out = sharded.trace_slab_torch(sc, opt, cam, ni, nj, 0, nj, details=details, counters=ctr)
    out["ctr"] = ctr
    return out
jobs = [(scene_variant("ks_ref0"), 320, 256), (scene_variant("ks_true0998_disk"), 256, 320)]
opt = rt.solver_defaults()
serial = []
for (sc, cam), ni, nj in jobs:
    serial.append(tr(sc, opt, cam, ni, nj)); torch.cuda.synchronize()
# serial again: deterministic?
for k, ((sc, cam), ni, nj) in enumerate(jobs):
    b = tr(sc, opt, cam, ni, nj); torch.cuda.synchronize()
    for key in serial[k]:
        x, y = serial[k][key], b[key]
        bad = ~((x == y) | (x.isnan() & y.isnan())) if x.is_floating_point() else (x != y)
        if bad.any(): print("serial repeat differs", k, key, int(bad.sum()))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("same_stream_pair", "two_streams"):
    for rep in range(4):
        outs = []
        for ((sc, cam), ni, nj), st in zip(jobs, (s1, s2 if mode == "two_streams" else s1)):
            outs.append(tr(sc, opt, cam, ni, nj, stream=st))
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(serial, outs)):
            for key in a:
                x, y = a[key], b[key]
                bad = ~((x == y) | (x.isnan() & y.isnan())) if x.is_floating_point() else (x != y)
                if bad.any():
                    idx = bad.nonzero()[:5].tolist()
                    print(mode, "rep", rep, "job", k, key, "n_bad", int(bad.sum()), "of", bad.numel(), idx)
print("done")
