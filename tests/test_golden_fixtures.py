"""Committed fp64 fixtures (tests/golden/oracle_*_32.npz, made by tests/golden/make_golden.py from the pinned oracle).

CPU: the oracle still reproduces them (guards against accidental changes of the oracle).
GPU: the HIP path matches them without needing the oracle library."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from scenes import rt, scene_variant

VARIANTS = ["mink", "ks_ref0", "ks_ref08", "ks_true0", "ks_true08", "ks_true0998", "ks_true0998_disk",
            "ks_ref0_shapes", "ks_true08_shapes", "mink_shapes",   # *_shapes: user-defined Object subtypes (examples/user_objects.py)
            "ks_ref0_many64", "ks_true08_many64"]                  # 64 objects: a list beyond the 16 inline slots (src/RayTraceGR.jl:433-441)
N = 32


def _fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", f"oracle_{name}_{N}.npz"))


@pytest.mark.parametrize("name", VARIANTS)
def test_oracle_reproduces_its_fixtures(name):
    import oracle_lib as O
    sc, cam = scene_variant(name, units=False)
    r = O.trace(sc, rt.solver_defaults(), N, N, cam=cam)
    f = _fixture(name)
    for k in ("status", "hit", "n_accept", "n_reject"):
        assert np.array_equal(r[k], f[k]), k
    assert np.allclose(r["state_end"], f["state_end"], rtol=1e-12, atol=1e-12)
    assert np.allclose(r["rgb"], f["rgb"], rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", VARIANTS)
def test_hip_path_matches_committed_fixtures(name):
    abi = rt._abi
    lib = abi.load()
    sc, cam = scene_variant(name)
    opt = rt.solver_defaults()
    n = N * N
    rgb = np.zeros((3, n))
    hit = np.zeros(n, np.uint8)
    status = np.zeros(n, np.uint8)
    nacc = np.zeros(n, np.uint32)
    nrej = np.zeros(n, np.uint32)
    o = abi.rtgr_ray_outputs()
    o.hit, o.status, o.n_accept, o.n_reject = hit.ctypes.data, status.ctypes.data, nacc.ctypes.data, nrej.ctypes.data
    abi.check(lib, lib.rtgr_trace_f64(None, C.byref(sc), C.byref(opt), None, C.byref(cam), N, N, 0, N, rgb.ctypes.data,
                                      C.byref(o), None))
    f = _fixture(name)
    flips = hit != f["hit"]
    assert int(flips.sum()) <= (40 if name.startswith("mink") else 2)   # Minkowski: noise-driven silhouettes
    same = ~flips
    assert (status[same] == f["status"][same]).all()
    nobj = sc.nobj
    d = np.abs(rgb[:, same] - f["rgb"][:, same])
    per = np.where(hit[same] > 0, hit[same] / max(nobj, 1), 1.0)[None, :]   # sawtooth period of a coloured hit
    e = np.minimum(d, np.abs(per - d)).max(axis=0)
    # (flat space, non-convex shapes: a ray whose samples step over the near side of a tube ends on its far side — same object,
    #  another colour; counted with the flips the Minkowski scenes are allowed)
    assert int((e > 1e-6).sum()) <= (max(0, 40 - int(flips.sum())) if name == "mink_shapes" else 0), np.sort(e)[-3:]
    steps_g = nacc.astype(np.int64) + nrej
    steps_f = f["n_accept"].astype(np.int64) + f["n_reject"]
    # (the *_many64 scenes' captured rays circle the hole for ~1300 steps until the far plane at t = -25: counts within 1.5 %)
    assert (np.abs(steps_g - steps_f) <= np.maximum(2, steps_f * 15 // 1000))[same].all()
