#!/usr/bin/env python3
"""Generates the fp64 golden fixtures tests/golden/oracle_<variant>_32.npz from the CPU oracle ("golden
vectors the build must generate itself").  The oracle is pinned on the reference's sphere.png / sphere2.png first
(tests/test_oracle_golden.py); these fixtures then freeze its full-precision outputs — end states, λ_end, RGB, hit
class, step counts — for a 32 x 32 render of every BASELINE.json scene variant, so the GPU parity tests have committed
vectors that do not depend on the oracle library being present.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from scenes import scene_variant, rt  # noqa: E402

VARIANTS = ["mink", "ks_ref0", "ks_ref08", "ks_true0", "ks_true08", "ks_true0998", "ks_true0998_disk"]
SHAPES = ["ks_ref0_shapes", "ks_true08_shapes", "mink_shapes"]   # user-defined Object subtypes (examples/user_objects.py; oracle twins)
MANY = ["ks_ref0_many64", "ks_true08_many64"]                    # an object list beyond the 16 inline slots (rtgr_scene.objects)
N = 32

if __name__ == "__main__":
    for name in (sys.argv[1:] or VARIANTS + SHAPES + MANY):
        sc, cam = scene_variant(name, units=False)
        r = O.trace(sc, rt.solver_defaults(), N, N, cam=cam)
        out = os.path.join(HERE, f"oracle_{name}_{N}.npz")
        np.savez_compressed(out, rgb=r["rgb"], state_end=r["state_end"], lambda_end=r["lambda_end"],
                            status=r["status"], hit=r["hit"], n_accept=r["n_accept"], n_reject=r["n_reject"])
        print(out, os.path.getsize(out), "bytes", "hit classes", np.bincount(r["hit"], minlength=4))
