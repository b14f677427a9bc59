"""Race hunt: the schedule of the persistent passes is non-deterministic (atomic queue, early list, LDS-buffered appends),
the results must not be.  Traces the same frame N times and compares every output with the first pass, bit for bit."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rt = load_package()
from raytracegr_jl_amd import sharded
from scenes import scene_variant
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for name in sys.argv[3:] or ["ks_ref0", "ks_true0998_disk"]:
    sc, cam = scene_variant(name)
    opt = rt.solver_defaults()
    ref, bad = None, 0
    for r in range(reps):
        ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
        out = sharded.trace_slab_torch(sc, opt, cam, n, n, 0, n, details=True, counters=ctr)
        torch.cuda.synchronize()
        cur = {k: v.clone() for k, v in out.items()}
        cur["counters"] = ctr[:7].clone()
        if ref is None:
            ref = cur
            continue
        for k in ref:
            a, b = ref[k], cur[k]
            same = torch.equal(a, b) if not a.is_floating_point() else bool(((a == b) | (a.isnan() & b.isnan())).all())
            if not same:
                bad += 1
                print(f"{name}: pass {r} differs in {k}")
    print(f"{name} {n}x{n}: {reps} passes, {bad} differences; counters {ref['counters'].tolist()}")
    assert bad == 0
