#!/usr/bin/env python3
"""One-off extension of the random-scene differential campaign (tests/test_campaign_random_scenes.py runs seeds 15-214 under pytest):
further seeds of the same generator, closed-form path and generic dual-number path, HIP against the oracle, with the stated bounds of
the random-scene test.  A script, not a test.      usage: python tests/campaign_builtin_objects.py [first_seed] [n_seeds]   -> stdout"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
from scenes import rt  # noqa: E402
from test_gpu_parity import _random_scene, hip_trace, random_scene_violations  # noqa: E402

abi = rt._abi
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 215
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
outside = {False: [], True: []}
for seed in range(first, first + count):
    sc0, cam, opt, nobj = _random_scene(seed)
    ref = O.trace(sc0, opt, 40, 32, cam=cam)
    for generic in (False, True):
        sc, _, _, _ = _random_scene(seed)
        if generic and sc.metric != abi.MINKOWSKI:
            sc.metric |= abi.METRIC_GENERIC
        gpu = hip_trace(lib, sc, opt, 40, 32, cam=cam)
        v = random_scene_violations(gpu, ref, sc0, nobj)
        if v:
            outside[generic].append(seed)
            print(f"seed {seed} ({'generic' if generic else 'closed'}; metric {sc0.metric}, M {sc0.M:.2f}, a {sc0.a:.2f}, reltol {opt.reltol:.1e}, "
                  f"lambda1 {opt.lambda1:.0f}, unfinished {int((ref['status'] >= 2).sum())}, max steps {int((ref['n_accept'] + ref['n_reject']).max())}): {v}", flush=True)
    if (seed - first + 1) % 50 == 0:
        print(f"… {seed - first + 1} scenes", flush=True)
print(f"seeds {first}..{first + count - 1}: {count} scenes x 2 paths; outside a bound — closed: {outside[False]}, generic: {outside[True]}")
