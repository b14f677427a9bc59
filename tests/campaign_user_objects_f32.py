#!/usr/bin/env python3
"""Float32 companion of tests/campaign_user_objects.py (a script, not a test): the same seeded scenes with user objects, traced in
Float32 by the HIP path (a unit's Float32 kernels: the single FULL pass) and by the Float32 oracle.  Float32 parity is statistical
(tests/test_gpu_parity.py: tolerance eps^(3/4) = 6.4e-6, steps ten times a Float64 one): per scene the fraction of pixels that end on
another object or differ by more than the Float32 RGB bar, and the ratio of step attempts.  A wrong kernel shows as a scene far out.
    usage: python tests/campaign_user_objects_f32.py [first_seed] [n_seeds]   -> stdout"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
from scenes import circular_channels, rt  # noqa: E402
from test_gpu_parity import F32_RGB_TOL, hip_trace  # noqa: E402
from test_user_objects import _random_shapes_scene  # noqa: E402

abi = rt._abi
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fracs, ratios, far_out = [], [], []
for seed in range(first, first + count):
    sc, cam, opt64, nobj = _random_shapes_scene(seed)
    opt = rt.solver_defaults(np.float32, lambda1=opt64.lambda1, hit_threshold=opt64.hit_threshold, miss_rgb=tuple(opt64.miss_rgb), max_steps=opt64.max_steps)
    gpu = hip_trace(lib, sc, opt, 40, 32, cam=cam, dtype=np.float32)
    ref = O.trace(sc, opt, 40, 32, cam=cam, dtype=np.float32)
    d = np.abs(gpu["rgb"].astype(float) - ref["rgb"].astype(float))
    per = np.where(gpu["hit"] > 0, gpu["hit"] / max(nobj, 1), 1.0)[None, :]
    e = np.where(circular_channels(gpu["hit"], sc), np.minimum(d, np.abs(per - d)), d).max(axis=0)
    off = (gpu["hit"] != ref["hit"]) | (gpu["status"] != ref["status"]) | (e >= F32_RGB_TOL)
    g = gpu["counters"]["accepted"] + gpu["counters"]["rejected"]
    r = ref["counters"]["accepted"] + ref["counters"]["rejected"]
    fracs.append(float(off.mean()))
    ratios.append(g / max(r, 1))
    if off.mean() > 0.05 or not 0.8 <= g / max(r, 1) <= 1.25:
        far_out.append(seed)
        print(f"seed {seed}: {off.mean():.3f} of the pixels off (hit {int((gpu['hit'] != ref['hit']).sum())}, status {int((gpu['status'] != ref['status']).sum())}), steps x {g / max(r, 1):.3f}", flush=True)
fr = np.array(fracs)
print(f"Float32, seeds {first}..{first + count - 1}: pixels off per scene — median {np.median(fr):.4f}, 90th percentile {np.percentile(fr, 90):.4f}, max {fr.max():.4f}; "
      f"step attempts HIP / oracle — {min(ratios):.3f} … {max(ratios):.3f}; scenes with > 5 % off or steps outside 0.8 … 1.25: {far_out}")
