#!/usr/bin/env python3
"""The random-scene differential campaign for USER objects (a script, not a test: ~5 min on a GPU box; tests/test_user_objects.py runs
seeds 0-11 of the same generator under pytest).  Seeded scenes that mix the torus and the ellipsoid of examples/user_objects.py with
planes, spheres and disks under four Kerr–Schild variants, traced by the HIP path and by the oracle, held to the stated bounds of the
random-scene test (tests/test_gpu_parity.py: random_scene_violations) — and every scene through rtgr_scene_check (FAR + NEAR against
the single FULL pass, bit for bit).      usage: python tests/campaign_user_objects.py [first_seed] [n_seeds]   -> stdout"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402
from scenes import rt  # noqa: E402
from test_gpu_parity import hip_trace, random_scene_violations  # noqa: E402
from test_user_objects import _random_shapes_scene  # noqa: E402

abi = rt._abi
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
outside, check_failed, rays, hits_user = [], [], 0, 0
for seed in range(first, first + count):
    sc, cam, opt, nobj = _random_shapes_scene(seed)
    gpu = hip_trace(lib, sc, opt, 40, 32, cam=cam)
    ref = O.trace(sc, opt, 40, 32, cam=cam)
    v = random_scene_violations(gpu, ref, sc, nobj)
    rays += 40 * 32
    user = [k + 1 for k in range(nobj) if sc.obj[k].kind == abi.USER_OBJECT]
    hits_user += int(sum((gpu["hit"] == k).sum() for k in user))
    if v:
        outside.append(seed)
        print(f"seed {seed}: outside the bounds: {v}", flush=True)
    if lib.rtgr_scene_check(None, C.byref(sc), C.byref(opt), C.byref(cam), 40, 32, 0) != 0:
        check_failed.append(seed)
        print(f"seed {seed}: scene check: {lib.rtgr_last_error().decode()}", flush=True)
    if (seed - first + 1) % 25 == 0:
        print(f"… {seed - first + 1} scenes", flush=True)
print(f"seeds {first}..{first + count - 1}: {count} scenes, {rays} rays, {hits_user} of them end on a user object; "
      f"{count - len(outside)} inside every bound, outside: {outside}; scene check (FAR + NEAR == FULL, bit for bit) failed on: {check_failed}")
