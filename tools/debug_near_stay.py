import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
rt = load_package()
from raytracegr_jl_amd import sharded
metric, objs, cam = rt.example2_scene()
sc, opt, camera = rt.make_scene(metric, objs), rt.solver_defaults(), rt.make_camera(**cam)
for n in (1024, 4096):
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    a = sharded.trace_slab_torch(sc, opt, camera, n, n, 0, n, details=True, counters=ctr)
    torch.cuda.synchronize()
    print(n, "max NEAR stay (accepted steps):", int(ctr[7]))
