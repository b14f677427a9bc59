"""Pins the CPU oracle (oracle/rtgr_oracle.cpp) on the reference's own committed outputs and on the hand-derived
known answers of SURVEY.md §4.2.  CPU only.

Golden provenance: tests/golden/sphere.png, sphere2.png are byte copies of /root/reference/sphere.png, sphere2.png
(= scenes/sphere.png, scenes/sphere2.png; md5 540cd312…, c40664eb…), the images example1()/example2() wrote
(src/RayTraceGR.jl:572-575, :608-611).  They are the only numeric goldens the reference holds for trace_rays.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT
from scenes import example, rt

GOLD = os.path.join(ROOT, "tests", "golden")


def _png(name):
    from raytracegr_jl_amd.png import read_png  # registered by conftest.load_package()
    return read_png(os.path.join(GOLD, name))


def test_golden_files_are_the_reference_outputs():
    md5 = lambda f: hashlib.md5(open(os.path.join(GOLD, f), "rb").read()).hexdigest()
    assert md5("sphere.png") == "540cd31208e2b23dfb18ae152beeafd7"
    assert md5("sphere2.png") == "c40664ebd4ffaf1f62c4feafc8bb9e4d"


@pytest.fixture(scope="module")
def ex2():
    sc, cam = example(2)
    return O.trace(sc, rt.solver_defaults(), 200, 200, cam=cam)


@pytest.fixture(scope="module")
def ex1():
    sc, cam = example(1)
    return O.trace(sc, rt.solver_defaults(), 200, 200, cam=cam)


def test_example2_matches_sphere2_png_exactly(ex2):
    """The acceptance test for the restated third-party semantics (Tsit5 + PI + callback): 40000/40000."""
    img = O.image_u8(ex2["rgb"], 200, 200)
    gold = _png("sphere2.png")
    assert gold.shape == (200, 200, 3)
    assert int((img != gold).any(axis=2).sum()) == 0
    # hit classes and workload statistics quoted in SURVEY §0.5 / §6
    assert np.bincount(ex2["hit"], minlength=4).tolist() == [0, 31338, 5154, 3508]
    steps = ex2["n_accept"] + ex2["n_reject"]
    assert int(steps.max()) == 991 and abs(steps.mean() - 210.5) < 0.05 and int(ex2["n_reject"].sum()) == 0
    assert ex2["counters"]["events_interior"] == 6
    assert (ex2["status"] == 0).all()


def test_example1_matches_sphere_png_outside_silhouette(ex1):
    """Minkowski: the embedded error estimate is rounding noise, so the silhouette ring of the small sphere is
    decided by floating-point noise in the reference.  Everything else must be exact."""
    img = O.image_u8(ex1["rgb"], 200, 200)
    gold = _png("sphere.png")
    bad = (img != gold).any(axis=2)
    assert int(bad.sum()) <= 150  # SURVEY: 145
    # every mismatch lies on the sphere/sky class boundary of the golden image (blue channel 255 = sphere hit)
    sph = gold[:, :, 2] == 255
    edge = np.zeros_like(sph)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            edge |= np.roll(np.roll(sph, dy, 0), dx, 1) != sph
    assert not (bad & ~edge).any()
    assert abs(int((ex1["hit"] == 3).sum()) - 3499) <= 60  # golden sphere pixel count 3499


KAT_PIXELS = [  # (i, j) 1-based, λ_end, step attempts, object, x_end, rgb
    ((1, 100), 5.057437336682, 977, 2, (-20, 1.18451444, 1.01755431, -0.00760096), (0, 1 / 3, 0)),
    ((30, 100), 8.117502705281, 792, 2, (-20, -0.60503554, 1.44003040, -0.00335175), (0, 1 / 3, 0)),
    ((100, 100), 2.163067184345, 27, 3, (-1.69514713, 4.02997684, -0.49906056, -0.00632041),
     (0.0482856, 0.2291619, 1)),
    ((150, 40), 14.71323869909, 95, 1, (-10.40973772, 5.92997685, 7.45720028, -3.03735716),
     (0.0596026, 0.1446273, 1 / 3)),
    ((200, 200), 12.22153494301, 75, 1, (-8.69088555, 7.74713066, 4.97198755, 3.90657219),
     (0.1556578, 0.0598159, 1 / 3)),
]


@pytest.mark.parametrize("ij,lam,nsteps,obj,xend,rgb", KAT_PIXELS)
def test_example2_known_pixels(ex2, ij, lam, nsteps, obj, xend, rgb):
    idx = (ij[0] - 1) + (ij[1] - 1) * 200
    assert abs(ex2["lambda_end"][idx] - lam) < 1e-11
    assert int(ex2["n_accept"][idx] + ex2["n_reject"][idx]) == nsteps
    assert int(ex2["hit"][idx]) == obj
    assert np.allclose(ex2["state_end"][idx, :4], xend, atol=2e-8)
    assert np.allclose(ex2["rgb"][:, idx], rgb, atol=2e-7)


def test_metric_known_answers():
    """KS as written at x=(0,2,0,0): ρ=2, r=3, f=2/3, k=(1,2/3,0,0)  (SURVEY §4.2)."""
    sc = rt.make_scene(rt.kerr_schild, [])
    g, dg, G = O.eval_metric(sc, [0, 2, 0, 0])
    g, dg, G = g[0], dg[0], G[0]
    t, x, y, z = 0, 1, 2, 3
    assert np.allclose([g[t, t], g[t, x], g[x, x], g[y, y], g[z, z]], [-1 / 3, 4 / 9, 35 / 27, 1, 1], atol=1e-15)
    assert np.allclose([dg[t, t, x], dg[t, x, x], dg[t, y, y], dg[t, z, z], dg[x, x, x], dg[x, y, y], dg[x, z, z]],
                       [-5 / 9, -14 / 27, 2 / 9, 2 / 9, -4 / 9, 4 / 27, 4 / 27], atol=1e-15)
    assert np.allclose([G[t, t, t], G[t, t, x], G[x, t, t], G[x, x, x], G[x, y, y]],
                       [0.19607843137254902, 0.5718954248366013, 0.14705882352941177, -0.4836601307189542,
                        0.2352941176470588], atol=1e-15)


RHS_KATS = [
    (("ref", 0.0), (-0.01289947892662181, -0.02993514609509498, 0.01496757304754749, -0.00224513595713212)),
    (("ref", 0.8), (-0.01497039373498526, -0.03389571946606656, 0.01614981228992642, -0.00280541310480339)),
    (("true", 0.8), (-0.02513041973653935, -0.06119090953648158, 0.02638759421409448, -0.00711246353919855)),
    (("true", 0.998), (-0.03042962226488852, -0.06334351525634142, 0.02515697417245203, -0.00818277300235165)),
]


@pytest.mark.parametrize("variant,udot", RHS_KATS)
def test_rhs_known_answers(variant, udot):
    m = rt.KerrSchild(1.0, variant[1], textbook=(variant[0] == "true"))
    sc = rt.make_scene(m, [])
    s = [0, 4, -2, 0.3, -1, 0.1, 0.7, 0.2]
    ds = O.geodesic(sc, s)[0]
    assert np.array_equal(ds[:4], s[4:])
    assert np.allclose(ds[4:], udot, rtol=0, atol=2e-16 + 1e-14 * np.abs(udot).max())
    # long-double evaluation of the same chain agrees: the double result is rounding-clean
    assert np.allclose(O.geodesic(sc, s, long_double=True)[0], ds, atol=2e-16)


def test_dual_jacobian_matches_finite_differences():
    rng = np.random.default_rng(7)
    for m in (rt.kerr_schild, rt.KerrSchild(1, 0.8, False), rt.KerrSchild(1, 0.8), rt.KerrSchild(1.3, 0.998)):
        sc = rt.make_scene(m, [])
        for _ in range(5):
            x = np.concatenate([[rng.normal()], rng.normal(size=3) * 2 + np.array([3, 0, 0])])
            g, dg, _ = O.eval_metric(sc, x)
            h = 1e-6
            for c in range(1, 4):
                e = np.zeros(4)
                e[c] = h
                fd = (O.metric_plain(sc, x + e)[0] - O.metric_plain(sc, x - e)[0]) / (2 * h)
                assert np.allclose(dg[0][:, :, c], fd, atol=5e-9)
            assert np.allclose(dg[0][:, :, 0], 0)
            assert np.allclose(g[0], O.metric_plain(sc, x)[0], atol=1e-15)
