#!/usr/bin/env python3
"""One-off differential campaign: the seeded random scenes of tests/test_gpu_parity.py (_random_scene) for MANY seeds,
through the closed-form path and the generic path, against the oracle with the test's own bounds.  Prints the seeds that
violate a bound and a summary.  It lives under tests/ (not collected by pytest) because it calls the oracle, which only test
code may do.      python tests/campaign_random_scenes.py [first_seed] [last_seed]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402

lib = T.abi.load()
T.abi.check(lib, lib.rtgr_init(-1))
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (15, 215)
bad = []
for seed in range(a, b):
    for generic in (False, True):
        try:
            if generic:
                orig = T._random_scene

                def gen(s, orig=orig):
                    sc, cam, opt, nobj = orig(s)
                    if sc.metric != T.abi.MINKOWSKI:
                        sc.metric |= T.abi.METRIC_GENERIC
                    return sc, cam, opt, nobj
                T._random_scene = gen
                try:
                    T.test_random_scenes_match_oracle(lib, seed)
                finally:
                    T._random_scene = orig
            else:
                T.test_random_scenes_match_oracle(lib, seed)
        except AssertionError:
            bad.append((seed, generic))
            print("seed", seed, "generic" if generic else "closed", "VIOLATES:", traceback.format_exc().splitlines()[-1][:300], flush=True)
print(f"{b - a} seeds x 2 paths: {len(bad)} violations {bad}")
