"""The reference's own unit tests (test/runtests.jl:12-61), restated against the CPU oracle (CPU) — the GPU twins
of the same checks are in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import oracle_lib as O
from scenes import rt


def test_minkowski_metric():
    """@testset "Minkowski metric" (test/runtests.jl:12-32).  The reference uses Rational{BigInt}; η has integer
    entries so Float64 is exact here too."""
    sc = rt.make_scene(rt.minkowski, [])
    x = np.zeros(4)
    g = O.metric_plain(sc, x)[0]
    gu = O.inv4(g)
    assert np.linalg.det(g) * np.linalg.det(gu) == 1          # :23
    assert np.array_equal(g @ gu, np.eye(4))                  # :24
    g1, dg, Gam = O.eval_metric(sc, x)
    assert np.array_equal(g1[0], g)                           # :27
    assert (dg == 0).all()                                    # :28
    assert (Gam == 0).all()                                   # :31


@pytest.mark.parametrize("i", range(1, 8))
def test_kerr_schild_metric(i):
    """@testset "Kerr-Schild metric" for i in 1:7, T = Float32 (test/runtests.jl:36-61)."""
    T = np.float32
    tol = np.finfo(T).eps ** T(0.75)                          # :38
    ix, iy, iz = i & 1, i & 2, i & 4
    x = np.array([0, 2 * ix, 2 * iy, 2 * iz], T)              # :44
    sc = rt.make_scene(rt.kerr_schild, [])
    g1, dg, Gam = O.eval_metric(sc, x, dtype=T)
    g = g1[0]
    assert not np.isnan(g).any()                              # :47
    gu = O.inv4(g.astype(np.float64)).astype(T)
    detg, detgu = np.linalg.det(g.astype(np.float64)), np.linalg.det(gu.astype(np.float64))
    assert abs(detg * detgu - 1) <= tol                       # :53
    assert np.abs(g.astype(np.float64) @ gu - np.eye(4)).max() <= tol  # :54
    assert np.abs(g - g1[0]).max() <= tol                     # :57 (vacuous in the reference too)
    assert not np.isnan(Gam).any()                            # :60


def test_rays_miss_colour():
    """The commented-out @testset "rays" (test/runtests.jl:65-79): no objects => miss colour (1,0,0)."""
    sc = rt.make_scene(rt.minkowski, [])
    opt = rt.solver_defaults(np.float32)
    s0 = np.array([[0, 0, 0, 0, -1, 1, 0, 0]], np.float32)
    r = O.trace(sc, opt, 1, 1, state0=s0, dtype=np.float32)
    assert np.abs(r["rgb"][:, 0] - [1, 0, 0]).max() <= np.finfo(np.float32).eps ** 0.75
    assert r["status"][0] == rt._abi.RAY_LAMBDA1
