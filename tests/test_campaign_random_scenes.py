"""The random-scene differential campaign (DESIGN §2 "where the bounds stop holding") under pytest: the seeded scene
generator of tests/test_gpu_parity.py for seeds 15-214, through the closed-form path AND the generic dual-number path,
against the oracle with the random-scene test's own bounds.

193 of the 200 seeds are inside every bound on both paths.  The 7 others are listed here by name.  The claim about them
(round 2: a script's printout, now an assertion) is that they are properties of the ALGORITHM — error estimates that are
rounding noise of the particular RHS formulation — not device errors:

  * five Minkowski scenes with several small spheres (seeds 25, 40, 90, 140, 170): Γ ≡ 0 makes the embedded error estimate
    pure rounding noise, steps grow to dt ≈ 2-10, and whether a ray "hits" a sphere narrower than the 10-sample spacing is
    decided by where the samples fall;
  * two Kerr–Schild scenes (56, 171) with a late plane and reltol 1e-9: captured rays reach |u^t| ~ 1e7 before they end,
    where the estimate is dominated by the rounding noise of the RHS formulation.

What is asserted for each of them: an INDEPENDENT device formulation of the same algorithm — the tile kernel (option
tile = 1: f64 controller, IEEE division, inline event finder; shares no integrate code with the production pipeline) —
is at least as far from the oracle, on the same bound, as the production kernels are (up to the factor 2 one expects
between two draws of the same noise).  A production-kernel defect would show as production alone being far out.  (How far
the two device formulations are from EACH OTHER is printed, not asserted: in Minkowski both evaluate the same noise — Γ ≡ 0,
the estimate is dt·Σb̃·u in both — and agree with each other while the oracle's 8-vector stage form rounds differently;
in the two Kerr–Schild scenes they differ from each other by a quarter of what either differs from the oracle.)"""
import numpy as np
import pytest

import oracle_lib as O
from scenes import rt
from test_gpu_parity import _random_scene, hip_trace, lib, random_scene_violations  # noqa: F401  (lib: fixture)

pytestmark = pytest.mark.gpu
abi = rt._abi
KNOWN_NOISE_DOMINATED = {25, 40, 56, 90, 140, 170, 171}
SEEDS = range(15, 215)


def _traces(lib, seed, generic):
    sc, cam, opt, nobj = _random_scene(seed)
    if generic and sc.metric != abi.MINKOWSKI:
        sc.metric |= abi.METRIC_GENERIC
    gpu = hip_trace(lib, sc, opt, 40, 32, cam=cam)
    sc0, _, _, _ = _random_scene(seed)
    ref = O.trace(sc0, opt, 40, 32, cam=cam)
    return sc0, cam, opt, nobj, gpu, ref


@pytest.mark.parametrize("generic", [False, True], ids=["closed", "generic"])
@pytest.mark.parametrize("seed", [s for s in SEEDS if s not in KNOWN_NOISE_DOMINATED])
def test_campaign_seed_is_inside_every_bound(lib, seed, generic):
    sc, cam, opt, nobj, gpu, ref = _traces(lib, seed, generic)
    v = random_scene_violations(gpu, ref, sc, nobj)
    assert not v, v


@pytest.mark.parametrize("generic", [False, True], ids=["closed", "generic"])
@pytest.mark.parametrize("seed", sorted(KNOWN_NOISE_DOMINATED))
def test_campaign_outlier_is_formulation_noise_not_a_device_error(lib, seed, generic):
    sc, cam, opt, nobj, prod, ref = _traces(lib, seed, generic)
    with abi.options(lib, tile=1):
        tile = hip_trace(lib, sc, opt, 40, 32, cam=cam)
    v_prod = random_scene_violations(prod, ref, sc, nobj)
    v_tile = random_scene_violations(tile, ref, sc, nobj)
    v_pair = random_scene_violations(prod, tile, sc, nobj)       # device formulation against device formulation
    print(f"seed {seed} {'generic' if generic else 'closed'}: production vs oracle {v_prod}; tile kernel vs oracle {v_tile}; "
          f"production vs tile kernel {v_pair}")
    assert v_prod, "this seed is listed as noise-dominated but the production path is inside every bound: unlist it"
    worst = max(v_prod, key=v_prod.get)
    assert worst in v_tile, (worst, v_prod, v_tile)                 # the independent formulation breaks the SAME bound ...
    assert v_tile[worst] >= 0.5 * v_prod[worst], (v_prod, v_tile)   # ... by at least as much (two draws of one noise: factor 2)


@pytest.mark.parametrize("seed", [246, 451])
def test_where_parity_ends_and_what_the_generic_path_holds_there(lib, seed):
    """INTEGRATION.md "Where parity ends", under test.  Two of the 14 outlier scenes of the 400-seed extension (profiles/r05/builtin_campaign.log,
    gpurun_out/r05/outliers.log): Kerr–Schild, a late plane, reltol 1e-9 — rays that are CAPTURED hover above the horizon with
    |u^t| growing to 1e7 and beyond before the plane ends them, and from |u^t| ~ 1e3 on the embedded error estimate is the rounding noise
    of whichever formulation of the RHS produced it.  What holds there, measured and pinned:
      * on the rays that stay signal-dominated (|u^t| < 1e3 at their end in the oracle) BOTH device formulations keep the library's
        parity: same status, step counts within +-3 of the oracle's;
      * on the noise-dominated rays the three formulations order as their cancellation predicts — oracle's as-written duals (noisiest:
        most steps) >= device generic duals >= device closed form (fewest) — and the GENERIC path, the reference's own formulation and
        the parity-first setting (`generic = true`), stays within 10 % of the oracle's mean step count where the closed form is up to 35 %
        below it; end positions and colours of the rays all three finish still agree to 1e-6."""
    sc0, cam, opt, nobj, closed, ref = _traces(lib, seed, False)
    _, _, _, _, generic, _ = _traces(lib, seed, True)
    steps = {k: (v["n_accept"] + v["n_reject"]).astype(np.int64) for k, v in (("oracle", ref), ("closed", closed), ("generic", generic))}
    ut = np.abs(ref["state_end"][:, 4])
    signal = ut < 1e3
    noisy = ~signal
    assert signal.sum() > 100 and noisy.sum() > 100, (int(signal.sum()), int(noisy.sum()))
    for name, got in (("closed", closed), ("generic", generic)):
        same_hit = got["hit"] == ref["hit"]
        ok = signal & same_hit
        assert (got["status"][ok] == ref["status"][ok]).all(), name
        assert np.abs(steps[name] - steps["oracle"])[ok].max() <= 3, (name, int(np.abs(steps[name] - steps["oracle"])[ok].max()))
        assert (signal & ~same_hit).sum() <= 6, name
    mean = {k: float(v[noisy].mean()) for k, v in steps.items()}
    print(f"seed {seed}: noise-dominated rays {int(noisy.sum())}, mean step attempts {mean}; at the step cap: oracle "
          f"{int((ref['status'] == 2).sum())}, generic {int((generic['status'] == 2).sum())}, closed {int((closed['status'] == 2).sum())}")
    assert mean["oracle"] >= mean["generic"] >= mean["closed"], mean
    assert mean["generic"] >= 0.90 * mean["oracle"], mean                      # the parity-first setting: within 10 % where it is all noise
    assert 0.60 * mean["oracle"] <= mean["closed"] < 0.97 * mean["oracle"], mean   # the closed form: fewer steps, as far as 35 % below
    done = noisy & (ref["status"] == 0) & (generic["status"] == 0) & (closed["status"] == 0) & (ref["hit"] == generic["hit"]) & (ref["hit"] == closed["hit"])
    assert done.sum() > 50
    from scenes import wrap_aware_rgb_err
    for got in (closed, generic):
        assert wrap_aware_rgb_err(got["rgb"][:, done], ref["rgb"][:, done], ref["hit"][done], sc=sc0) <= 1e-6
