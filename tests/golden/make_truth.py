#!/usr/bin/env python3
"""Generates tests/golden/truth_<variant>.npz: for 64 pixels of a 200 x 200 render of every BASELINE.json scene variant,
the end state, λ_end, hit object and RGB of the TRUE geodesic as tests/truth.py integrates it (sympy-differentiated
metric, scipy DOP853 at rtol 1e-13 — no code or algorithm shared with the oracle or the HIP path), plus

  self_err   |end state(rtol 1e-13) − end state(rtol 1e-11)|∞ — the truth's own convergence (the error of the LOOSER run)
  clearance  smallest |distance| to any object other than the one hit, along the ray (small = grazing; see truth.py)

Pixels: one per cell of an 8 x 8 grid over the image, jittered with a fixed seed (no look at any solver's output).

    python tests/golden/make_truth.py            # ≈ 2 min on 8 cores
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

VARIANTS = ["mink", "ks_ref0", "ks_ref08", "ks_true0", "ks_true08", "ks_true0998", "ks_true0998_disk", "schw_iso", "kerr_bl",
            "ks_ref0_shapes", "ks_true08_shapes"]   # *_shapes: user-defined Object subtypes (examples/user_objects.py)
N, GRID = 200, 8


def pixels():
    rng = np.random.default_rng(20261004)
    cell = N // GRID
    ij = [(gi * cell + int(rng.integers(cell)), gj * cell + int(rng.integers(cell))) for gj in range(GRID) for gi in range(GRID)]
    return np.array(ij, dtype=np.int64)


def truth_scene(name):
    """scene_variant(name), or — 'schw_iso' — example2's objects and camera around an isotropic-coordinates Schwarzschild hole
    (the user-metric example; only kind / M enter here, no module is compiled or loaded)."""
    from scenes import scene_variant, rt
    if name not in ("schw_iso", "kerr_bl"):
        return scene_variant(name, units=False)
    sc, cam = scene_variant("ks_true0" if name == "schw_iso" else "ks_true08")
    sc.metric = rt._abi.USER
    if name == "kerr_bl":          # Kerr a = 0.8 in Boyer–Lindquist coordinates (examples/user_metrics.py)
        import truth
        sc.user_metric = truth.KERR_BL_MARKER
        assert sc.obj[0].kind == rt._abi.SPHERE and sc.obj[0].p[8] == -10.0
        sc.obj[0].p[8] = -8.0      # sky sphere at r = 8: Boyer–Lindquist time runs ~1.5 x faster along these rays than
        #                            Kerr–Schild time, and example2's plane (t = −20) would cut most rays to r = 10 short
    return sc, cam


def one(job):
    import truth
    from scenes import scene_variant, rt
    name, i, j = job
    sc, cam = truth_scene(name)
    opt = rt.solver_defaults()
    s0 = truth.pixel_state(sc, cam, N, N, i, j)
    r = truth.trace_ray(sc, opt, s0)
    loose = truth.trace_ray(sc, opt, s0, rtol=1e-11, atol=1e-13)
    self_err = np.abs(r["state_end"] - loose["state_end"]).max() if loose["hit"] == r["hit"] else np.inf
    return s0, r["state_end"], r["lambda_end"], r["hit"], r["rgb"], r["clearance"], self_err, r["nfev"]


if __name__ == "__main__":
    ij = pixels()
    only = sys.argv[1:]          # e.g. `make_truth.py schw_iso`: regenerate just these
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        for name in (only or VARIANTS):
            res = pool.map(one, [(name, int(i), int(j)) for i, j in ij], chunksize=2)
            out = os.path.join(HERE, f"truth_{name}.npz")
            extra = {}
            if name == "kerr_bl":   # no oracle twin exists for this metric: the pointwise g(x) and RHS vectors come from here too
                import truth
                sc, _ = truth_scene(name)
                rng = np.random.default_rng(9)
                s = np.concatenate([rng.uniform(-6, 6, (400, 4)), rng.uniform(-1, 1, (400, 4))], axis=1)
                s = s[(np.linalg.norm(s[:, 1:4], axis=1) > 2.5) & (np.hypot(s[:, 1], s[:, 2]) > 0.5)][:128]
                f = truth.rhs(sc)
                extra = dict(rhs_states=s, rhs_values=np.array([f(0.0, v) for v in s]),
                             metric_values=np.array([truth.metric(sc, v[:4]) for v in s]))
            np.savez_compressed(out, n=N, ij=ij, **extra, state0=np.array([r[0] for r in res]),
                                state_end=np.array([r[1] for r in res]), lambda_end=np.array([r[2] for r in res]),
                                hit=np.array([r[3] for r in res], dtype=np.uint8), rgb=np.array([r[4] for r in res]),
                                clearance=np.array([r[5] for r in res]), self_err=np.array([r[6] for r in res]),
                                nfev=np.array([r[7] for r in res]))
            se = np.array([r[6] for r in res])
            print(name, os.path.getsize(out), "bytes; hit classes", np.bincount([r[3] for r in res], minlength=4),
                  "self_err median %.1e max %.1e" % (np.median(se), se.max()), flush=True)
