"""User-defined objects on the device (RTGR_USER_OBJECT; rtgr_user_unit_compile / api.UserObjects).

The reference's `Object{T}` is an OPEN abstract type: any subtype with `distance(obj, pos)` and `objcolor(obj, pos)` is traced —
the ContinuousCallback condition dispatches on it (src/RayTraceGR.jl:433-441) and so does the colour rule (:518-530); `Plane` and
`Sphere` are merely the two the file ships (:374-428).  The product compiles new subtypes, given as device source, into the scene's
run-time unit together with the metric they are traced with.

CPU part: the build rules, the units' symbols on both build routes, the oracle's twins against their committed fixtures.
GPU part: scenes with the torus / ellipsoid of examples/user_objects.py beside built-in objects against the oracle (and, in
tests/test_truth.py, against independent true geodesics), Float64 and Float32, hipcc-built and in-process, FAR + NEAR bit-identical
to FULL, a unit that carries no reach bound, a unit that carries a metric too, every entry point, and the refusals."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT
from scenes import rt, scene_variant, user_shapes

sys.path.insert(0, os.path.join(ROOT, "examples"))
import user_metrics  # noqa: E402
import user_objects  # noqa: E402

abi = rt._abi
um = sys.modules[rt.__name__ + ".user_metric"]
UNIT_KERNELS = ["rtgr_user_integrate_far", "rtgr_user_integrate_near", "rtgr_user_integrate_full10", "rtgr_user_integrate_fulln",
                "rtgr_user_prepare", "rtgr_user_resolve", "rtgr_user_integrate_full10_f32", "rtgr_user_integrate_fulln_f32",
                "rtgr_user_prepare_f32", "rtgr_user_resolve_f32", "rtgr_user_eval_objects", "rtgr_user_eval_objects_f32"]
UNIT_GLOBALS = ["rtgr_user_abi_version", "rtgr_user_header_hash", "rtgr_user_unit_desc", "rtgr_user_far_waves", "rtgr_user_near_waves",
                "rtgr_user_f32_waves"]
METRIC_ONLY = ["rtgr_user_canvas", "rtgr_user_eval_metric", "rtgr_user_eval_geodesic", "rtgr_user_eval_accel"]


def _symbols(path):
    syms = subprocess.run([os.path.join(um.LLVM_BIN, "llvm-readelf"), "--dyn-syms", "-W", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in syms.splitlines() if line.strip()}


# ---- CPU -----------------------------------------------------------------------------------------------------------------
def test_what_a_unit_is_made_of_is_read_off_its_source():
    d, st = um.unit_defines(user_objects.SHAPES_WITH_REACH, built_for=(abi.KS_REF, False, False))
    assert d == ["-DRTGR_UNIT_BUILTIN_METRIC=1", "-DRTGR_UNIT_GENERIC=0", "-DRTGR_UNIT_SPIN=0", "-DRTGR_USER_OBJECTS=1", "-DRTGR_USER_REACH=1"] and not st
    d, _ = um.unit_defines(user_objects.SHAPES, built_for=(abi.KS_TRUE, True, True))
    assert d == ["-DRTGR_UNIT_BUILTIN_METRIC=2", "-DRTGR_UNIT_GENERIC=1", "-DRTGR_UNIT_SPIN=0", "-DRTGR_USER_OBJECTS=1"]   # generic: one variant
    d, _ = um.unit_defines(user_objects.SHAPES, built_for=(abi.MINKOWSKI, True, True))
    assert d[:3] == ["-DRTGR_UNIT_BUILTIN_METRIC=0", "-DRTGR_UNIT_GENERIC=0", "-DRTGR_UNIT_SPIN=0"]                       # Minkowski: one variant
    d, st = um.unit_defines(user_metrics.KERR_SCHILD_KS + user_objects.SHAPES_WITH_REACH)
    assert d == ["-DRTGR_USER_NE=3", "-DRTGR_USER_KS=1", "-DRTGR_USER_OBJECTS=1", "-DRTGR_USER_REACH=1"] and st
    assert um.unit_defines(user_metrics.SCHWARZSCHILD_ISOTROPIC) == ([], False)                                           # a metric alone: as before
    with pytest.raises(ValueError, match="built for ONE built-in metric"):
        um.unit_defines(user_objects.SHAPES)
    with pytest.raises(ValueError, match="both methods"):
        um.unit_defines(user_objects.SHAPES.replace("rtgr_user_objcolor", "rtgr_user_paint"), built_for=(1, False, False))
    with pytest.raises(ValueError, match="built_for must be None"):
        um.unit_defines(user_metrics.KERR_SCHILD_KS + user_objects.SHAPES, built_for=(1, False, False))
    with pytest.raises(ValueError, match="rtgr_user_reach without"):
        um.unit_defines(user_metrics.KERR_SCHILD_KS + user_objects.REACH)
    with pytest.raises(ValueError):
        rt.UserObjects(user_metrics.KERR_SCHILD_KS + user_objects.SHAPES)
    with pytest.raises(ValueError):
        rt.UserObjects("int nothing_here;")


def test_a_comment_that_mentions_a_function_does_not_define_it(tmp_path):
    """What a unit is made of is read off the source's CODE — an identifier followed by `(`, outside comments and string literals — by
    the Python mirror (`defines`) and by the library (`source_defines`, through rtgr_user_unit_build) alike.  A comment that mentions
    rtgr_user_reach used to switch -DRTGR_USER_REACH=1 on (the build then failed on an undefined template); one that mentions
    rtgr_user_metric made an objects-only source a "metric" unit.  … and rtgr_user_sample is detected the same way."""
    chatty = ("// a torus and an egg; no rtgr_user_reach( bound ) here, and nothing like rtgr_user_metric(x) either\n"
              "/* rtgr_user_ks(x, M, a, f, k) would be another way to give a metric */\n" + user_objects.SHAPES +
              '\n// const char* note = "rtgr_user_sample(type, p)";\n')
    d, st = um.unit_defines(chatty, built_for=(abi.KS_REF, False, False))
    assert d == ["-DRTGR_UNIT_BUILTIN_METRIC=1", "-DRTGR_UNIT_GENERIC=0", "-DRTGR_UNIT_SPIN=0", "-DRTGR_USER_OBJECTS=1"] and not st
    assert um.defines(user_objects.SHAPES_WITH_REACH_AND_SAMPLES, "rtgr_user_sample") and not um.defines(chatty, "rtgr_user_sample")
    d, _ = um.unit_defines(user_objects.SHAPES_WITH_REACH_AND_SAMPLES, built_for=(abi.KS_REF, False, False))
    assert d[-3:] == ["-DRTGR_USER_OBJECTS=1", "-DRTGR_USER_REACH=1", "-DRTGR_USER_SAMPLE=1"]
    rt.UserObjects(chatty, name="chatty", jit=True)         # (constructing it checks the same thing: objects only, both methods)
    # the library's own reading of the same text: built_for is REQUIRED (objects only) and the unit builds — no reach bound switched on
    lib = abi.load()
    sc = rt.make_scene(rt.kerr_schild, [], units=False)
    out = str(tmp_path / "chatty.hsaco")
    assert lib.rtgr_user_unit_build(chatty.encode(), 0, None, out.encode()) == abi.ERR_BAD_ARG and b"built_for" in lib.rtgr_last_error()
    abi.check(lib, lib.rtgr_user_unit_build(chatty.encode(), 0, C.byref(sc), out.encode()))
    syms = _symbols(out)
    assert "rtgr_user_resolve" in syms and "rtgr_user_samples" not in syms
    out2 = str(tmp_path / "samples.hsaco")
    abi.check(lib, lib.rtgr_user_unit_build(user_objects.SHAPES_WITH_REACH_AND_SAMPLES.encode(), 0, C.byref(sc), out2.encode()))
    assert "rtgr_user_samples" in _symbols(out2)


def test_pruning_the_unit_cache_touches_units_only(tmp_path, monkeypatch):
    """prune_stale_units decides by NAME and content digest: a cache directory shared with other files (RTGR_USER_CACHE=/tmp …) keeps
    them, a unit of the current device headers survives a `touch` of the headers' mtimes, and units named after another state of the
    headers — or in the naming of earlier rounds — go (the rule used to be "every file older than the newest header")."""
    import os
    import time
    monkeypatch.setenv("RTGR_USER_CACHE", str(tmp_path))
    cur = um._env_digest()
    keep = [tmp_path / "notes.txt", tmp_path / "metric_notes.hsaco", tmp_path / f"metric_{cur}_{'a' * 20}.hsaco", tmp_path / f"metric_{cur}_{'b' * 20}.hip"]
    gone = [tmp_path / f"metric_{'0' * 8}_{'a' * 20}.hsaco", tmp_path / f"metric_{'c' * 20}.hsaco", tmp_path / f"metric_{'0' * 8}_{'d' * 20}.L1.tmp123.s"]
    for f in keep + gone:
        f.write_text("x")
        os.utime(f, (time.time() - 10 ** 7, time.time() - 10 ** 7))          # all of them OLDER than every header
    assert um.prune_stale_units() == len(gone)
    assert all(f.exists() for f in keep) and not any(f.exists() for f in gone)


@pytest.mark.parametrize("route", ["hipcc", "in-process"])
def test_object_units_build_on_cpu_with_every_kernel(route, tmp_path):
    """A unit of objects for a built-in metric variant carries the integrate / set-up / RESOLVE kernels (Float64 and Float32) and none
    of the kernels only a metric of its own needs; a unit with both carries all of them.  No spills, audit clean."""
    for name, src, st, bf, with_metric in (("ksref0", user_objects.SHAPES_WITH_REACH, False, (abi.KS_REF, False, False), False),
                                           ("kstrue spin, no reach", user_objects.SHAPES, False, (abi.KS_TRUE, False, True), False),
                                           ("own metric + objects", user_metrics.KERR_SCHILD_KS + user_objects.SHAPES_WITH_REACH, True, None, True)):
        path = (um.compile_user_metric(src, stationary=st, built_for=bf) if route == "hipcc"
                else um.build_in_process(src, str(tmp_path / (name.replace(" ", "_").replace(",", "") + ".hsaco")), stationary=st, built_for=bf))
        names = _symbols(path)
        for k in UNIT_KERNELS + UNIT_GLOBALS:
            assert k in names, (name, k)
        for k in METRIC_ONLY:
            assert (k in names) == with_metric, (name, k)
        scratch = um.code_object_scratch(path)
        assert len(scratch) == 6 and max(scratch.values()) <= um.MAX_SCRATCH, (name, scratch)
        assert um.audit(path) == (0, ""), name
    lib = abi.load()
    out = str(tmp_path / "bad.hsaco")
    assert lib.rtgr_user_unit_build(user_objects.SHAPES.encode(), 0, None, out.encode()) == abi.ERR_BAD_ARG     # objects alone need built_for
    assert b"built_for" in lib.rtgr_last_error() and not os.path.exists(out)
    sc = abi.rtgr_scene()
    sc.metric = abi.USER
    assert lib.rtgr_user_unit_build(user_objects.SHAPES.encode(), 0, C.byref(sc), out.encode()) == abi.ERR_BAD_ARG
    sc.metric = abi.KS_REF
    assert lib.rtgr_user_unit_build(user_metrics.KERR_SCHILD_KS.encode(), 1, C.byref(sc), out.encode()) == abi.ERR_BAD_ARG   # own metric + built-in built_for
    bad = user_objects.SHAPES.replace("msqrt", "no_such_function")
    assert lib.rtgr_user_unit_build(bad.encode(), 0, C.byref(sc), out.encode()) == abi.ERR_BAD_ARG
    assert b"no_such_function" in lib.rtgr_last_error() and not os.path.exists(out)


def test_scene_description_without_units_needs_neither_compiler_nor_gpu():
    sc, _ = scene_variant("ks_true08_shapes", units=False)
    assert sc.nobj == 5 and sc.user_metric == 0 and [sc.obj[k].kind for k in range(5)] == [abi.SPHERE, abi.PLANE] + [abi.USER_OBJECT] * 3
    assert [sc.obj[k].type for k in range(5)] == [0, 0, user_objects.TORUS, user_objects.ELLIPSOID, user_objects.TORUS]
    assert list(sc.obj[3].p)[:6] == [3.3, 1.0, -0.8, 0.7, 0.5, 0.5]
    fam_a, fam_b = rt.UserObjects(user_objects.SHAPES), rt.UserObjects(user_objects.SHAPES)
    with pytest.raises(ValueError, match="number of object types"):          # (two families: joined — which needs their type counts)
        rt.make_scene(rt.kerr_schild, [fam_a(0, [1, 2, 3, 1, 0.1]), fam_b(1, [0] * 6)], units=False)


def _two_families(reach=(True, True), jit=False):
    """Objects of TWO sources in one scene: the shapes of examples/user_objects.py and the reference's Sphere written as a user
    object -> (joined scene objects after [sky, plane], the same with the built-in Sphere in place of the user one)"""
    key = (tuple(reach), jit)
    if key not in _TWO:
        _TWO[key] = (rt.UserObjects(user_objects.SHAPES_WITH_REACH if reach[0] else user_objects.SHAPES, name="shapes", ntypes=2, jit=jit),
                     rt.UserObjects(user_objects.SPHERE_AS_USER_OBJECT if reach[1] else user_objects.SPHERE_AS_USER_OBJECT.split(
                         "template <class S> __device__ S rtgr_user_reach")[0], name="sphere", ntypes=1, jit=jit))
    shapes, ball = _TWO[key]
    pos, vel, radius = [0.0, 4.6, -0.9, 0.9], [1.0, 0.0, 0.0, 0.0], 0.35
    joined = [shapes(user_objects.TORUS, [4.0, 0.0, 0.0, 0.9, 0.3]), ball(0, pos + vel + [radius]), shapes(user_objects.ELLIPSOID, [3.3, 1.0, -0.8, 0.7, 0.5, 0.5])]
    return joined, [joined[0], rt.Sphere(pos, vel, radius), joined[2]]


_TWO = {}


def test_sources_of_several_families_join_into_one():
    """rtgr_user_source_join: text in, text out (no GPU) — a namespace per family, dispatchers on the renumbered tag; make_scene
    joins by itself and adds each family's base to its objects' types, in the order of first appearance."""
    lib = abi.load()
    joined, mixed = _two_families()
    metric, objs, _ = rt.example2_scene()
    sc = rt.make_scene(metric, objs[:2] + joined, units=False)
    assert [sc.obj[k].kind for k in range(5)] == [abi.SPHERE, abi.PLANE] + [abi.USER_OBJECT] * 3
    assert [sc.obj[k].type for k in range(5)] == [0, 0, 0, 2, 1]             # shapes: 0, 1; the sphere family's type 0 -> 2
    sc = rt.make_scene(metric, objs[:2] + joined[1:], units=False)            # the sphere's family first: it takes base 0
    assert [sc.obj[k].type for k in range(4)] == [0, 0, 0, 2]
    fam, bases = rt.UserObjects.join([joined[0].family, joined[1].family])
    assert bases == [0, 2] and fam.ntypes == 3 and rt.UserObjects.join([joined[0].family, joined[1].family])[0] is fam
    assert fam.source.count("namespace rtgr_family_") == 2 and "rtgr_family_1::rtgr_user_reach(type - 2u, x, p, dl)" in fam.source
    assert um.unit_defines(fam.source, False, (abi.KS_REF, False, False))[0][-2:] == ["-DRTGR_USER_OBJECTS=1", "-DRTGR_USER_REACH=1"]
    # one family without a bound: its types are never provably out of reach; none with a bound: no reach function at all
    half = rt.UserObjects.join([_two_families((False, True))[0][0].family, _two_families((False, True))[0][1].family])[0]
    assert "if (type < 2u) return S(__builtin_huge_val());" in half.source and "rtgr_family_1::rtgr_user_reach" in half.source
    none = rt.UserObjects.join([_two_families((False, False))[0][0].family, _two_families((False, False))[0][1].family])[0]
    assert "rtgr_user_reach" not in none.source
    with pytest.raises(ValueError, match="defines the object types 0..1"):
        joined[0].family(2, [0] * 5)
    # the C entry point's own refusals
    need = C.c_uint64(0)

    def join(srcs, nts, buf=None, cap=0):
        a = (C.c_char_p * len(srcs))(*[None if t is None else t.encode() for t in srcs])
        return lib.rtgr_user_source_join(a, (C.c_uint32 * len(nts))(*nts), len(srcs), buf, cap, C.byref(need))
    assert join([user_objects.SHAPES, user_objects.SPHERE_AS_USER_OBJECT], [2, 1]) == 0 and need.value == len(half.source.encode()) + 1
    small = C.create_string_buffer(16)
    for bad, why in (((lambda: join([user_objects.SHAPES], [0])), "number of object types is 0"),
                     ((lambda: join([user_metrics.KERR_BOYER_LINDQUIST], [1])), "must define rtgr_user_distance"),
                     ((lambda: join([user_metrics.KERR_BOYER_LINDQUIST + user_objects.SHAPES], [2])), "defines a metric"),
                     ((lambda: join([fam.source, user_objects.SHAPES], [3, 2])), "joined source itself"),
                     ((lambda: join([user_objects.SHAPES, None], [2, 1])), "source 1 is NULL"),
                     ((lambda: join([user_objects.SHAPES], [2], small, 16)), "the buffer holds 16 bytes")):
        assert bad() == abi.ERR_BAD_ARG and why in lib.rtgr_last_error().decode(), (why, lib.rtgr_last_error())
    assert lib.rtgr_user_source_join(None, None, 0, None, 0, None) == abi.ERR_BAD_ARG
    # a mistake in ONE family's text is reported with that family's name and its own line number (`#line` per namespace)
    broken = rt.UserObjects(user_objects.SPHERE_AS_USER_OBJECT.replace("macos", "no_such_function"), name="broken", ntypes=1)
    both = rt.UserObjects.join([joined[0].family, broken])[0]
    at = 1 + broken.source.split("\n").index(next(l for l in broken.source.split("\n") if "no_such_function" in l))
    with pytest.raises(RuntimeError, match="no_such_function") as e:
        um.compile_user_metric(both.source, built_for=(abi.KS_REF, False, False))
    assert f"object family 1:{at}:" in str(e.value), str(e.value)[:800]


def test_oracle_twins_obey_the_distance_contract():
    """`zero on the surface, positive outside, negative inside` (src/RayTraceGR.jl:377-383), through the oracle's trace of flat-space
    rays that start inside and outside each shape: both end with an event ON the surface (the condition's initial sign is negative
    for the one and positive for the other)."""
    _, (torus, egg, _) = user_shapes()
    opt = rt.solver_defaults()
    d_torus = lambda x: (np.hypot(x[0] - 4.0, x[1]) - 0.9) ** 2 + x[2] ** 2 - 0.3 ** 2
    d_egg = lambda x: ((x[0] - 3.3) / 0.7) ** 2 + ((x[1] - 1.0) / 0.5) ** 2 + ((x[2] + 0.8) / 0.5) ** 2 - 1.0
    cases = ((torus, d_torus, (4.9, 0.0, 0.0), (0.0, 0.0, 1.0)),      # from the tube's centre line upwards: leaves the tube at z = 0.3
             (torus, d_torus, (4.0, 0.0, 0.0), (1.0, 0.0, 0.0)),      # from the hole's centre outwards: meets the tube at x = 4.6
             (egg, d_egg, (3.3, 1.0, -0.8), (0.0, 0.0, -1.0)),        # from the centre downwards
             (egg, d_egg, (3.3, 1.0, 0.2), (0.0, 0.0, -1.0)))         # from above: meets the top at z = -0.3
    for obj, dist, start, direction in cases:
        sc = rt.make_scene(rt.minkowski, [obj], units=False)
        s0 = np.array([[0.0, *start, -1.0, *direction]])
        r = O.trace(sc, opt, 1, 1, state0=s0)
        assert r["hit"][0] == 1 and r["status"][0] == abi.RAY_EVENT, (start, r["hit"], r["status"])
        assert abs(dist(r["state_end"][0, 1:4])) < 1e-12 and np.sign(dist(np.array(start))) in (-1.0, 1.0)
    assert d_torus(np.array((4.9, 0.0, 0.0))) < 0 < d_torus(np.array((4.0, 0.0, 0.0))) and d_egg(np.array((3.3, 1.0, -0.8))) < 0 < d_egg(np.array((3.3, 1.0, 0.2)))


# ---- GPU -----------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def lib():
    lib = abi.load()
    abi.check(lib, lib.rtgr_init(-1))
    return lib


def _info(sc):
    return um.unit_info(sc.user_metric)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ks_ref0_shapes", "ks_true08_shapes", "mink_shapes", "ks_ref08_shapes", "ks_ref0_generic_shapes"])
def test_scenes_with_user_objects_match_the_oracle(lib, name):
    """The north_star bar for the last argument of trace_rays(metric, objs, canvas): hit map equal to the oracle's, RGB within 1e-6
    (wrap-aware on the channels the shapes put a sawtooth on), statuses equal, step counts within ±1 — and the FAR + NEAR passes (the
    source gives a reach bound) bit-identical to the single FULL pass, as for the built-in objects."""
    from test_gpu_parity import compare, hip_trace
    sc, cam = scene_variant(name)
    info = _info(sc)
    assert info["has_objects"] == 1 and info["has_reach"] == 1 and info["probe_ok"] == 1
    assert info["metric"] == (sc.metric if (sc.metric & ~abi.METRIC_GENERIC) != abi.MINKOWSKI else abi.MINKOWSKI)
    assert info["spin"] == int(name in ("ks_true08_shapes", "ks_ref08_shapes", "ks_ref0_generic_shapes"))
    opt = rt.solver_defaults()
    ref = O.trace(sc, opt, 64, 64, cam=cam)
    assert set(np.unique(ref["hit"])) >= {1, 3, 4, 5}                       # sky, torus, ellipsoid, second torus all on screen
    got = hip_trace(lib, sc, opt, 64, 64, cam=cam)
    if name == "mink_shapes":
        # Flat space: the embedded error estimate is rounding noise, steps grow to dt ~ 2-10 and whether the sampled sign test
        # (9 samples per step) sees a tube 0.4-0.6 across is decided by where the samples fall — between ANY two
        # implementations of the reference's algorithm a few per cent of such pixels end on another crossing (the tube's far side:
        # same object, other colour) or another object (tests/test_campaign_random_scenes.py).  Stated bound: 95 % of
        # the pixels agree in object AND colour at the 1e-6 bar; every ray is accounted for; nothing flies through the sky sphere.
        from scenes import circular_channels
        d = np.abs(got["rgb"] - ref["rgb"])
        per = np.where(got["hit"] > 0, got["hit"] / sc.nobj, 1.0)[None, :]
        e = np.where(circular_channels(got["hit"], sc), np.minimum(d, np.abs(per - d)), d).max(axis=0)
        agree = (got["hit"] == ref["hit"]) & (e <= 1e-6)
        assert agree.mean() >= 0.95, agree.mean()
        assert (got["status"] == 0).all() and (got["hit"] > 0).all() and got["counters"]["rays"] == 64 * 64
    else:
        compare(got, ref, max_class_flips=0, max_step_diff=1, sc=sc)
    with abi.options(lib, split=0):
        full = hip_trace(lib, sc, opt, 64, 64, cam=cam)
    for k in ("rgb", "hit", "status", "n_accept", "n_reject", "state_end", "lambda_end"):
        assert np.array_equal(got[k], full[k], equal_nan=got[k].dtype.kind == "f"), (name, k)
    assert got["counters"] == full["counters"]
    again = hip_trace(lib, sc, opt, 64, 64, cam=cam)
    assert np.array_equal(again["rgb"], got["rgb"]) and np.array_equal(again["n_accept"], got["n_accept"])


@pytest.mark.gpu
@pytest.mark.parametrize("metric_name", ["ks_ref0", "ks_true08"])
def test_user_objects_in_a_long_list_match_the_oracle(lib, metric_name):
    """New `Object` subtypes among the 60 objects of a long list: the unit's kernels walk the device table like the library's own —
    the spheres in groups, the user objects behind them with the other kinds, their reach bound asked per step and their distance
    bounds (distance ± reach over the event's step) in the resolve kernel's selection — and the colour rule sees the caller's order.
    Against the oracle at the north_star bar, and bit for bit against the pass structures that ask everything."""
    from scenes import many_objects, user_shapes
    from test_gpu_parity import compare, hip_trace
    metric = {"ks_ref0": rt.kerr_schild, "ks_true08": rt.KerrSchild(1, 0.8)}[metric_name]
    spheres = many_objects(57)                                   # sky sphere, far plane, 55 small spheres
    shapes = user_shapes(True)[1]
    objs = spheres[:20] + shapes[:1] + spheres[20:40] + shapes[1:] + spheres[40:]
    sc, cam = rt.make_scene(metric, objs), rt.make_camera(**rt.example2_scene()[2])
    assert sc.nobj == 60 and _info(sc)["has_reach"] == 1
    opt = rt.solver_defaults()
    ref = O.trace(sc, opt, 56, 56, cam=cam)
    user_hits = {21, 42, 43}                                     # the shapes' places in the caller's list (1-based)
    assert user_hits <= set(np.unique(ref["hit"])) and len(np.unique(ref["hit"])) > 12
    got = hip_trace(lib, sc, opt, 56, 56, cam=cam)
    compare(got, ref, sc=sc, max_class_flips=4, max_step_diff=2, rel_step_diff=0.015)
    for knobs in (dict(split=0), dict(groups=0)):
        with abi.options(lib, **knobs):
            other = hip_trace(lib, sc, opt, 56, 56, cam=cam)
        for k in ("rgb", "hit", "status", "n_accept", "n_reject", "state_end", "lambda_end"):
            assert np.array_equal(got[k], other[k], equal_nan=got[k].dtype.kind == "f"), (knobs, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ks_ref0_shapes", "ks_true08_shapes"])
def test_user_objects_in_float32_match_the_float32_oracle(lib, name):
    """T = Float32 (`Object{T}` is generic in T, src/RayTraceGR.jl:375): the unit's Float32 kernels against the Float32 oracle at the
    Float32 bars of tests/test_gpu_parity.py."""
    from test_gpu_parity import F32_FLIP_FRAC, F32_RGB_TOL, F32_STEPS, hip_trace
    sc, cam = scene_variant(name)
    opt = rt.solver_defaults(np.float32)
    gpu = hip_trace(lib, sc, opt, 64, 64, cam=cam, dtype=np.float32)
    ref = O.trace(sc, opt, 64, 64, cam=cam, dtype=np.float32)
    # (a torus is not convex: a Float32 step — ten times a Float64 one — can carry the sampled sign test over the near side of the
    #  tube and end the ray on the far side: same object, another colour.  Such pixels count as class flips do.)
    from scenes import circular_channels
    d = np.abs(gpu["rgb"].astype(float) - ref["rgb"].astype(float))
    per = np.where(gpu["hit"] > 0, gpu["hit"] / sc.nobj, 1.0)[None, :]
    e = np.where(circular_channels(gpu["hit"], sc), np.minimum(d, np.abs(per - d)), d).max(axis=0)
    off = (gpu["hit"] != ref["hit"]) | (e >= F32_RGB_TOL)
    assert off.mean() <= F32_FLIP_FRAC, (off.mean(), int((gpu["hit"] != ref["hit"]).sum()))
    g = gpu["counters"]["accepted"] + gpu["counters"]["rejected"]
    r = ref["counters"]["accepted"] + ref["counters"]["rejected"]
    assert F32_STEPS[0] * r <= g <= F32_STEPS[1] * r, (gpu["counters"], ref["counters"])
    assert gpu["counters"]["rays"] == 64 * 64


@pytest.mark.gpu
def test_a_unit_without_a_reach_bound_runs_the_full_pass_and_traces_the_same_frame(lib):
    """rtgr_user_reach is optional: without it no user object is ever provably out of reach, the unit's scenes run the single FULL
    pass (every accepted step scanned, as the reference does) and the frame is the one the FAR + NEAR passes of the unit WITH the
    bound deliver — the bound decides when a step is scanned, never what the scan finds."""
    from test_gpu_parity import compare, hip_trace
    sc_r, cam = scene_variant("ks_ref0_shapes")
    sc_n, _ = scene_variant("ks_ref0_shapes", reach=False)
    assert sc_r.user_metric != sc_n.user_metric and _info(sc_n)["has_reach"] == 0 and _info(sc_n)["probe_ok"] == 1
    opt = rt.solver_defaults()
    a, b = hip_trace(lib, sc_r, opt, 48, 40, cam=cam), hip_trace(lib, sc_n, opt, 48, 40, cam=cam)
    compare(b, O.trace(sc_n, opt, 48, 40, cam=cam), max_step_diff=1, sc=sc_n)
    for k in ("hit", "status", "n_accept", "n_reject"):
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(a["rgb"] - b["rgb"]).max() <= 1e-12 and np.abs(a["state_end"] - b["state_end"]).max() <= 1e-12
    lib.rtgr_timing_enable(None, 0, 1)
    hip_trace(lib, sc_n, opt, 48, 40, cam=cam)
    ms, n = (C.c_double * 4)(), (C.c_uint64 * 4)()
    abi.check(lib, lib.rtgr_timing_read(None, 0, C.byref(ms), C.byref(n)))
    lib.rtgr_timing_enable(None, 0, 0)
    assert n[1] >= 1 and n[3] == 0                                            # one main (FULL) pass, no NEAR pass


@pytest.mark.gpu
@pytest.mark.parametrize("metric_name", ["ks_ref0", "ks_true08"])
def test_objects_of_two_families_share_one_scene(lib, metric_name):
    """The reference puts ANY mix of Object subtypes into `objs` (src/RayTraceGR.jl:483, :518-530).  Objects of two separately written
    sources — the torus / ellipsoid family and the reference's own Sphere typed as a user object — in one scene: make_scene joins
    the sources into one unit.  Against the oracle (which has the torus and ellipsoid twins and the built-in Sphere) at the
    north-star bar; and, HIP against HIP, the SAME frame bit for bit as the scene with the built-in Sphere in the user sphere's
    place (the joined unit calls the very functions each family's own unit does), FAR + NEAR == FULL, Float64 and Float32."""
    from test_gpu_parity import compare, hip_trace
    metric = {"ks_ref0": rt.kerr_schild, "ks_true08": rt.KerrSchild(1, 0.8)}[metric_name]
    _, objs, cam = rt.example2_scene()
    cam = rt.make_camera(**cam)
    joined, mixed = _two_families()
    sc_j, sc_m = rt.make_scene(metric, objs[:2] + joined), rt.make_scene(metric, objs[:2] + mixed)
    assert sc_j.user_metric != sc_m.user_metric and [sc_j.obj[k].type for k in range(2, 5)] == [0, 2, 1]
    assert _info(sc_j) == dict(_info(sc_m), **{k: _info(sc_j)[k] for k in ("far_waves", "near_waves", "f32_waves")})
    assert _info(sc_j)["has_reach"] == 1 and _info(sc_j)["probe_ok"] == 1
    opt = rt.solver_defaults()
    ref = O.trace(rt.make_scene(metric, objs[:2] + mixed, units=False), opt, 64, 64, cam=cam)
    assert set(np.unique(ref["hit"])) >= {1, 3, 4, 5}
    got, same = hip_trace(lib, sc_j, opt, 64, 64, cam=cam), hip_trace(lib, sc_m, opt, 64, 64, cam=cam)
    compare(got, ref, max_class_flips=0, max_step_diff=1, sc=sc_m)
    with abi.options(lib, split=0):
        full = hip_trace(lib, sc_j, opt, 64, 64, cam=cam)
    for k in ("rgb", "hit", "status", "n_accept", "n_reject", "state_end", "lambda_end"):
        assert np.array_equal(got[k], same[k], equal_nan=got[k].dtype.kind == "f"), (metric_name, k, "joined unit vs built-in sphere")
        assert np.array_equal(got[k], full[k], equal_nan=got[k].dtype.kind == "f"), (metric_name, k, "FAR + NEAR vs FULL")
    abi.check(lib, lib.rtgr_scene_check(None, C.byref(sc_j), C.byref(opt), C.byref(cam), 48, 48, 0))
    opt32 = rt.solver_defaults(np.float32)
    a, b = (hip_trace(lib, sc, opt32, 64, 64, cam=cam, dtype=np.float32) for sc in (sc_j, sc_m))
    for k in ("rgb", "hit", "status", "n_accept", "n_reject"):
        assert np.array_equal(a[k], b[k]), (metric_name, k, "Float32")
    # pointwise: the joined dispatchers hand each type to its own family
    x = np.array([[0.0, 4.9, 0.0, 0.3], [0.0, 4.6, -0.9, 1.25], [0.0, 3.3, 1.0, -0.3], [0.0, 4.0, 0.0, 0.7]])
    dj, dm = rt.eval_objects(metric, objs[:2] + joined, x), rt.eval_objects(metric, objs[:2] + mixed, x)
    for k in ("d", "dmin", "hit", "rgb"):
        assert np.array_equal(dj[k], dm[k]), k
    assert list(dj["hit"]) == [3, 4, 5, 0]


@pytest.mark.gpu
def test_joined_families_build_in_process_too(lib):
    """The route a Julia or C caller takes (rtgr_user_unit_compile: hiprtc + comgr inside the library, no hipcc): the joined text —
    namespaces, `#line` directives, dispatchers — compiles there as well, and traces the frame of the hipcc-built unit."""
    from test_gpu_parity import hip_trace
    _, objs, cam = rt.example2_scene()
    cam = rt.make_camera(**cam)
    opt = rt.solver_defaults()
    sc_h = rt.make_scene(rt.kerr_schild, objs[:2] + _two_families()[0])
    sc_j = rt.make_scene(rt.kerr_schild, objs[:2] + _two_families(jit=True)[0])
    assert sc_h.user_metric != sc_j.user_metric and _info(sc_j)["has_reach"] == 1 and _info(sc_j)["probe_ok"] == 1
    a, b = hip_trace(lib, sc_h, opt, 48, 48, cam=cam), hip_trace(lib, sc_j, opt, 48, 48, cam=cam)
    for k in ("hit", "status", "n_accept", "n_reject"):
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(a["rgb"] - b["rgb"]).max() <= 1e-12          # (two compilers' fusions of the same source: not bit for bit)
    abi.check(lib, lib.rtgr_scene_check(None, C.byref(sc_j), C.byref(opt), C.byref(cam), 48, 48, 0))


@pytest.mark.gpu
def test_a_family_without_a_bound_beside_one_with_a_bound(lib):
    """Joined sources where only SOME bring rtgr_user_reach: the types of the others answer +infinity — a scene that holds such an
    object scans every step (all rays take the NEAR pass), a scene of the same unit without one keeps the FAR pass; the frames are
    those of the unit where every family has its bound."""
    from test_gpu_parity import hip_trace
    _, objs, cam = rt.example2_scene()
    cam = rt.make_camera(**cam)
    opt = rt.solver_defaults()
    both, half = _two_families()[0], _two_families((False, True))[0]
    sc_b, sc_h = rt.make_scene(rt.kerr_schild, objs[:2] + both), rt.make_scene(rt.kerr_schild, objs[:2] + half)
    assert sc_b.user_metric != sc_h.user_metric and _info(sc_h)["has_reach"] == 1
    a, b = hip_trace(lib, sc_b, opt, 48, 48, cam=cam), hip_trace(lib, sc_h, opt, 48, 48, cam=cam)
    for k in ("rgb", "hit", "status", "n_accept", "n_reject", "state_end", "lambda_end"):
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), k
    # only the sphere (the family WITH a bound) on the scene, through the half-bounded unit: same frame as through the fully bounded one
    sc_b1, sc_h1 = rt.make_scene(rt.kerr_schild, objs[:2] + both[1:2]), rt.make_scene(rt.kerr_schild, objs[:2] + half[1:2])
    assert sc_b1.user_metric not in (sc_b.user_metric, sc_h.user_metric)      # (one family alone: its own unit, not the joined one)
    a, b = hip_trace(lib, sc_b1, opt, 48, 48, cam=cam), hip_trace(lib, sc_h1, opt, 48, 48, cam=cam)
    assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["n_accept"], b["n_accept"])


@pytest.mark.gpu
def test_units_built_in_process_trace_the_same_frames(lib):
    """rtgr_user_unit_compile — one call from source text plus the scene the unit is meant for, for C and Julia callers without hipcc —
    against the oracle, and against the hipcc-built unit of the same source."""
    from test_gpu_parity import compare, hip_trace
    opt = rt.solver_defaults()
    for name in ("ks_ref0_shapes", "ks_true08_shapes"):
        sc_j, cam = scene_variant(name, jit=True)
        sc_h, _ = scene_variant(name)
        assert _info(sc_j) == {**_info(sc_h)}                                  # same variant, occupancies, probe
        got = hip_trace(lib, sc_j, opt, 48, 48, cam=cam)
        compare(got, O.trace(sc_j, opt, 48, 48, cam=cam), max_step_diff=1, sc=sc_j)
        ref = hip_trace(lib, sc_h, opt, 48, 48, cam=cam)
        assert np.array_equal(got["hit"], ref["hit"]) and np.abs(got["rgb"] - ref["rgb"]).max() <= 1e-9
    # the raw C entry: a scene, its source, one call
    sc, cam = scene_variant("ks_true0_shapes", units=False)
    mid = C.c_uint64(0)
    abi.check(lib, lib.rtgr_user_unit_compile(None, user_objects.SHAPES_WITH_REACH.encode(), 0, C.byref(sc), C.byref(mid)))
    sc.user_metric = mid.value
    compare(hip_trace(lib, sc, opt, 32, 32, cam=cam), O.trace(sc, opt, 32, 32, cam=cam), max_step_diff=1, sc=sc)


@pytest.mark.gpu
def test_a_unit_may_carry_a_metric_and_objects(lib):
    """Both extension points of trace_rays(metric, objs, canvas) at once: textbook Kerr–Schild typed as user source (Kerr–Schild form)
    and the shapes, compiled into one unit — against the oracle's built-in metric with its shape twins, at the user-metric bars."""
    from test_gpu_parity import compare, hip_trace
    user = rt.UserMetric(user_metrics.KERR_SCHILD_KS, M=1.0, a=0.8, name="ks as source")
    _, objs, cam = rt.example2_scene()
    fam, shapes = user_shapes()
    sc = rt.make_scene(user, objs[:2] + shapes)
    info = _info(sc)
    assert info["metric"] == abi.USER and info["has_objects"] == 1 and info["has_reach"] == 1 and info["probe_ok"] == 1
    sco, camera = scene_variant("ks_true08_shapes", units=False)
    opt = rt.solver_defaults()
    ref = O.trace(sco, opt, 48, 48, cam=camera)
    compare(hip_trace(lib, sc, opt, 48, 48, cam=camera), ref, max_class_flips=2, max_step_diff=2, sc=sc)
    with abi.options(lib, split=0):
        compare(hip_trace(lib, sc, opt, 48, 48, cam=camera), ref, max_class_flips=2, max_step_diff=2, sc=sc)
    # the unit's own canvas / hooks serve the metric as for a metric-only unit
    x = np.array([[0.0, 3.0, 1.0, 0.5]])
    g = rt.dmetric(user, x)[0]
    assert np.abs(g - rt.dmetric(rt.KerrSchild(1.0, 0.8), x)[0]).max() < 1e-14


@pytest.mark.gpu
def test_user_objects_through_every_entry_point(lib):
    """The reference's own call — trace_rays(metric, objs, canvas) with an Array{Pixel} — the single-ray shape, a strided share of
    the rows, and a context that lists the GPU twice: the same pixels as the device-resident frame."""
    from test_gpu_parity import hip_trace
    sc, cam = scene_variant("ks_ref0_shapes")
    metric, objs, camd = rt.example2_scene()
    objs = objs[:2] + user_shapes()[1]
    opt = rt.solver_defaults()
    ni = nj = 40
    frame = hip_trace(lib, sc, opt, ni, nj, cam=cam)
    canvas = rt.make_canvas(metric, camd["pos"], camd["widthx"], camd["widthy"], camd["normal"], ni, nj)
    out = rt.trace_rays(metric, objs, canvas)
    rgb = np.stack([p.reshape(-1, order="F") for p in out.rgb_planes()])
    assert np.array_equal(rgb, frame["rgb"])
    px = canvas.pixels[17, 22]
    one = rt.trace_ray(metric, objs, None, px)
    assert np.array_equal(one["rgb"], frame["rgb"][:, 17 + 22 * ni])
    ctx = abi.create_context(lib, [0, 0])
    try:
        out2 = rt.trace_rays(metric, objs, canvas, ctx=ctx)
        assert np.array_equal(np.stack([p.reshape(-1, order="F") for p in out2.rgb_planes()]), frame["rgb"])
    finally:
        lib.rtgr_destroy(ctx)


@pytest.mark.gpu
def test_scenes_never_run_with_the_wrong_unit(lib):
    from test_gpu_parity import hip_trace
    sc, cam = scene_variant("ks_ref0_shapes")
    opt = rt.solver_defaults()
    wrong = abi.rtgr_scene.from_buffer_copy(sc)
    wrong.a = 0.8                                            # the a != 0 instantiation is another kernel
    with pytest.raises(abi.RtgrError, match="another metric variant"):
        hip_trace(lib, wrong, opt, 8, 8, cam=cam)
    wrong = abi.rtgr_scene.from_buffer_copy(sc)
    wrong.metric = abi.KS_TRUE
    with pytest.raises(abi.RtgrError, match="another metric variant"):
        hip_trace(lib, wrong, opt, 8, 8, cam=cam)
    wrong = abi.rtgr_scene.from_buffer_copy(sc)
    wrong.user_metric = 0x1234
    with pytest.raises(abi.RtgrError, match="not loaded"):
        hip_trace(lib, wrong, opt, 8, 8, cam=cam)
    metric_only = rt.UserMetric(user_metrics.SCHWARZSCHILD_ISOTROPIC, stationary=True)
    wrong = abi.rtgr_scene.from_buffer_copy(sc)
    wrong.user_metric = metric_only.module_id()
    with pytest.raises(abi.RtgrError, match="defines no rtgr_user_distance"):
        hip_trace(lib, wrong, opt, 8, 8, cam=cam)
    wrong = abi.rtgr_scene.from_buffer_copy(sc)
    wrong.metric = abi.USER                                  # … and a unit of objects alone has no metric of its own
    with pytest.raises(abi.RtgrError, match="defines no metric"):
        hip_trace(lib, wrong, opt, 8, 8, cam=cam)
    with abi.options(lib, tile=1):
        with pytest.raises(abi.RtgrError, match="persistent pipeline"):
            hip_trace(lib, sc, opt, 8, 8, cam=cam)
    # a built-in scene is untouched by resident units, and an unknown kind is still refused
    plain, _ = scene_variant("ks_ref0")
    assert hip_trace(lib, plain, opt, 8, 8, cam=cam)["counters"]["rays"] == 64
    plain.obj[2].kind = 5
    with pytest.raises(abi.RtgrError, match="unknown object kind"):
        hip_trace(lib, plain, opt, 8, 8, cam=cam)


@pytest.mark.gpu
def test_a_reach_bound_that_lies_is_caught_by_the_scene_check(lib):
    """rtgr_user_reach is the one piece of a user's source the library cannot verify by construction: the FAR pass skips the scan of a
    step when no object can change sign within the bound, so a bound that is too small loses hits SILENTLY.  rtgr_scene_check traces
    the caller's own scene through the single FULL pass and through FAR + NEAR and compares: a sound bound passes bit for bit, a bound
    of zero ("nothing ever moves") is refused with the number of rays that differ, and so is a bound that forgets one object type.
    Round 6: the comparison runs BY ITSELF the first time a scene with such objects is traced (rtgr_units.hip: auto_scene_check) —
    the lying scene below is refused by a plain trace call, through the device entry and through the reference's own pixel array
    alike, with nobody calling check_scene; option scene_check = 0 gives the old behaviour back (and shows what is lost)."""
    _, objs, cam = rt.example2_scene()
    shapes = user_shapes()[1]
    rt.check_scene(rt.kerr_schild, objs[:2] + shapes, cam)                              # examples/user_objects.py REACH: sound
    rt.check_scene(rt.KerrSchild(1.0, 0.8), objs[:2] + shapes, cam)
    rt.check_scene(rt.kerr_schild, objs, cam)                                           # a built-in scene passes too (the library's own bounds)
    liar = rt.UserObjects(user_objects.SHAPES + """
template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]) { return S(0); }
""", name="shapes with a reach bound of zero")
    lying = objs[:2] + [liar(o.type, o.params) for o in shapes]
    with pytest.raises(abi.RtgrError, match=r"differ \(\d+ of 2304 rays.*rtgr_user_reach is not an upper bound"):
        rt.check_scene(rt.kerr_schild, lying, cam)
    half = rt.UserObjects(user_objects.SHAPES + user_objects.REACH.replace("if (type == 0u) {", "if (type == 0u) { return S(0);"),
                          name="shapes whose torus claims never to move")
    with pytest.raises(abi.RtgrError, match="rtgr_scene_check"):
        rt.check_scene(rt.kerr_schild, objs[:2] + [half(o.type, o.params) for o in shapes], cam)
    # WITHOUT the caller's help: the lying unit loads (its probe frame holds built-in objects: the source offers no samples), and the
    # first trace of the scene is refused by the automatic check — camera rays (device-side make_canvas) …
    from test_gpu_parity import hip_trace
    sc = rt.make_scene(rt.kerr_schild, lying)
    camera = rt.make_camera(**cam)
    opt = rt.solver_defaults()
    with pytest.raises(abi.RtgrError, match=r"differ \(\d+ of \d+ rays.*rtgr_user_reach is not an upper bound.*automatic check"):
        hip_trace(lib, sc, opt, 64, 64, cam=camera)
    with pytest.raises(abi.RtgrError, match="refused by the automatic scene check when it was first traced"):   # … the second time from the table
        hip_trace(lib, sc, opt, 64, 64, cam=camera)
    # … and the reference's own call shape, trace_rays(metric, objs, canvas): ray states from the caller's Pixel array
    canvas = rt.make_canvas(rt.kerr_schild, cam["pos"], cam["widthx"], cam["widthy"], cam["normal"], 64, 64)
    opt9 = rt.solver_defaults(reltol=1e-9, abstol=1e-9)         # (other solver constants: another scene as far as the table goes)
    with pytest.raises(abi.RtgrError, match="automatic check"):
        rt.trace_rays(rt.kerr_schild, lying, canvas, opt=opt9)
    # the sound scene goes through the same check and is traced (first call checked, second answered from the table)
    sound = rt.make_scene(rt.kerr_schild, objs[:2] + shapes)
    a = hip_trace(lib, sound, opt, 64, 64, cam=camera)
    b = hip_trace(lib, sound, opt, 64, 64, cam=camera)
    assert np.array_equal(a["rgb"], b["rgb"])
    # scene_check = 0: the lying unit traces — its FULL pass is right, its FAR + NEAR frame is not, which is why the check exists
    with abi.options(lib, scene_check=0):
        with abi.options(lib, split=0):
            full = hip_trace(lib, sc, opt, 48, 48, cam=camera)
        assert np.array_equal(full["hit"], O.trace(sc, opt, 48, 48, cam=camera)["hit"])
        assert (hip_trace(lib, sc, opt, 48, 48, cam=camera)["hit"] != full["hit"]).sum() > 0


@pytest.mark.gpu
def test_a_unit_whose_samples_expose_a_lying_bound_does_not_load(lib):
    """The optional fourth function of an object source, rtgr_user_sample, hands the load-time probe one object per type: the probe's
    FULL-against-FAR + NEAR comparison then runs through the unit's OWN distance and reach functions, and a bound that lies about a
    sample keeps the unit from loading at all.  The sound source with samples loads (probe_ok) and traces the oracle's frame."""
    _, objs, cam = rt.example2_scene()
    good = rt.UserObjects(user_objects.SHAPES_WITH_REACH_AND_SAMPLES, name="shapes + reach + samples", jit=True)
    sc = rt.make_scene(rt.kerr_schild, objs[:2] + [good(o.type, o.params) for o in user_shapes()[1]])
    info = um.unit_info(sc.user_metric)
    assert info["probe_ok"] == 1 and info["has_reach"] == 1
    from test_gpu_parity import hip_trace, compare
    camera, opt = rt.make_camera(**cam), rt.solver_defaults()
    compare(hip_trace(lib, sc, opt, 48, 48, cam=camera), O.trace(sc, opt, 48, 48, cam=camera), sc=sc, max_class_flips=2, max_step_diff=2)
    liar = rt.UserObjects(user_objects.SHAPES + """
template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]) { return S(0); }
""" + user_objects.SAMPLE, name="a reach bound of zero, with samples", jit=True)
    with pytest.raises(abi.RtgrError, match="refused by the load-time probe.*FULL pass and its FAR \\+ NEAR passes"):
        rt.make_scene(rt.kerr_schild, objs[:2] + [liar(o.type, o.params) for o in user_shapes()[1]])


def _random_shapes_scene(seed):
    """Seeded random scene with user objects among built-in ones: random Kerr–Schild variant (units for four metric variants), an
    optional sky sphere, 2-6 objects drawn from {plane, sphere, disk, torus, ellipsoid} in random order with random parameters, random
    camera and solver constants — the random-scene test of tests/test_gpu_parity.py with the two new Object subtypes in the mix."""
    rng = np.random.default_rng(1000 + seed)
    metric = [rt.kerr_schild, rt.KerrSchild(1.0, 0.6), rt.KerrSchild(0.7, 0.9, textbook=False), rt.KerrSchild(1.3, 0.0)][seed % 4]
    fam = user_shapes()[0]
    objs = [rt.Sphere((0, 0, 0, 0), (1, 0, 0, 0), -rng.uniform(9, 14))] if seed % 3 else []
    n_user = 0
    for k in range(rng.integers(2, 7)):
        kind = rng.integers(0, 6) if (k or n_user) else 4          # (at least one user object per scene)
        c = rng.normal(size=3) * 1.5 + np.array([4, 1, 0])
        if kind == 0:
            objs.append(rt.Plane(-rng.uniform(8, 30)))
        elif kind == 1:
            objs.append(rt.Sphere((0, *c), (1, 0, 0, 0), rng.uniform(0.2, 1.2)))
        elif kind == 2:
            objs.append(rt.Disk(rng.uniform(0.02, 0.2), rng.uniform(2.5, 4), rng.uniform(5, 8)))
        elif kind in (3, 4):
            objs.append(fam(user_objects.TORUS, [*c, rng.uniform(0.5, 1.4), rng.uniform(0.15, 0.4)]))
            n_user += 1
        else:
            objs.append(fam(user_objects.ELLIPSOID, [*c, *rng.uniform(0.25, 1.0, size=3)]))
            n_user += 1
    pos = (0, 4 + rng.normal(), -3 + rng.normal(), rng.normal() * 0.5)
    cam = rt.make_camera(pos, (0, 1, 0, 0), (0, 0, 0, 1), (0, 0.1 * rng.normal(), 1, 0.1 * rng.normal()))
    opt = rt.solver_defaults(lambda1=float(rng.choice([100.0, 15.0])), reltol=float(rng.choice([2.0 ** -39, 1e-9])),
                             hit_threshold=float(rng.choice([0.01, 0.05])), miss_rgb=(0.25, 0.5, 0.75), max_steps=5000)
    return rt.make_scene(metric, objs), cam, opt, len(objs)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_with_user_objects_match_the_oracle(lib, seed):
    """The stated bounds of test_random_scenes_match_oracle (tests/test_gpu_parity.py: unfinished rays by count, <= 6 class flips,
    <= 2 pixels over the 1e-6 wrap-aware RGB bar outside long orbits, step counts p99 <= 3) on scenes that mix the torus and the
    ellipsoid with planes, spheres and disks — and, the reach bound coming from the source, the scene check on each of them."""
    from test_gpu_parity import hip_trace, random_scene_violations
    sc, cam, opt, nobj = _random_shapes_scene(seed)
    gpu = hip_trace(lib, sc, opt, 40, 32, cam=cam)
    ref = O.trace(sc, opt, 40, 32, cam=cam)
    v = random_scene_violations(gpu, ref, sc, nobj)
    assert not v, v
    abi.check(lib, lib.rtgr_scene_check(None, C.byref(sc), C.byref(opt), C.byref(cam), 40, 32, 0))


@pytest.mark.gpu
def test_the_same_compile_call_again_is_answered_at_once(lib):
    """rtgr_user_unit_compile remembers, per context, what it built: the same (source, stationary, metric variant) again returns the
    id of the resident unit without a compiler run — a C or Julia caller need not keep a table of its own —; another variant is
    another build; after an unload the call builds again."""
    import time
    src = user_objects.SHAPES.replace("S(6)", "S(5)").encode()                 # (a source no other test compiles in-process)
    sc, _ = scene_variant("ks_ref0", units=False)
    ids, secs = [], []
    for _ in range(3):
        t0 = time.time()
        out = C.c_uint64(0)
        abi.check(lib, lib.rtgr_user_unit_compile(None, src, 0, C.byref(sc), C.byref(out)))
        ids.append(out.value)
        secs.append(time.time() - t0)
    assert ids[0] == ids[1] == ids[2] and secs[0] > 0.5 and max(secs[1:]) < 0.25 * secs[0], secs
    other, _ = scene_variant("ks_true08", units=False)
    out = C.c_uint64(0)
    abi.check(lib, lib.rtgr_user_unit_compile(None, src, 0, C.byref(other), C.byref(out)))
    assert out.value != ids[0]
    abi.check(lib, lib.rtgr_user_metric_unload(None, ids[0]))
    t0 = time.time()
    abi.check(lib, lib.rtgr_user_unit_compile(None, src, 0, C.byref(sc), C.byref(out)))
    assert out.value == ids[0] and time.time() - t0 > 0.5                      # not resident any more: built (and probed) again
    for i in (ids[0], ):
        abi.check(lib, lib.rtgr_user_metric_unload(None, i))


@pytest.mark.gpu
def test_in_process_units_can_be_kept_on_disk(lib, tmp_path, monkeypatch):
    """RTGR_UNIT_CACHE (opt-in): the code object rtgr_user_unit_compile builds is kept under a key of everything it depends on, and the
    same call in a later process — here: after unloading the unit — loads the file instead of compiling: same id, probed again, in a
    fraction of the time; another source or another metric variant is another file."""
    import time
    monkeypatch.setenv("RTGR_UNIT_CACHE", str(tmp_path))
    sc, cam = scene_variant("ks_true0_shapes", units=False)
    src = (user_objects.SHAPES_WITH_REACH + "\n// cache test\n").encode()

    def compile_once(scene):
        mid = C.c_uint64(0)
        t0 = time.perf_counter()
        abi.check(lib, lib.rtgr_user_unit_compile(None, src, 0, C.byref(scene), C.byref(mid)))
        return mid.value, time.perf_counter() - t0
    first, t_build = compile_once(sc)
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 1 and files[0].startswith("unit_") and files[0].endswith(".hsaco")
    abi.check(lib, lib.rtgr_user_metric_unload(None, first))
    again, t_load = compile_once(sc)
    assert again == first and um.unit_info(again)["probe_ok"] == 1 and t_load < 0.5 * t_build, (t_build, t_load)
    other = abi.rtgr_scene.from_buffer_copy(sc)
    other.a = 0.7                                                  # the a != 0 instantiation: another unit, another file
    third, _ = compile_once(other)
    assert third != first and len(os.listdir(tmp_path)) == 2
    sc.user_metric = again
    from test_gpu_parity import compare, hip_trace
    opt = rt.solver_defaults()
    compare(hip_trace(lib, sc, opt, 24, 24, cam=cam), O.trace(sc, opt, 24, 24, cam=cam), max_step_diff=1, sc=sc)


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", [("ks_ref0_shapes", 1024), ("ks_true08_shapes", 2048)])
def test_full_size_frames_with_user_objects_have_the_size_independent_properties(lib, name, size):
    """BASELINE-size frames of the scenes with user objects, where the oracle is out of reach: every ray accounted for and ended by an
    event, the hit-class fractions of the 64² frame that IS checked against the oracle (within 0.5 %), the same mean step count per ray,
    a slab of rows bit-equal to those rows of the full frame, and a 4-way cyclic share through the strided-rows entry too."""
    import torch
    from raytracegr_jl_amd import sharded
    from test_gpu_parity import hip_trace
    sc, cam = scene_variant(name)
    opt = rt.solver_defaults()
    small = hip_trace(lib, sc, opt, 64, 64, cam=cam)
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda")
    a = sharded.trace_slab_torch(sc, opt, cam, size, size, 0, size, details=True, counters=ctr)
    torch.cuda.synchronize()
    hit = a["hit"].cpu().numpy()
    frac, frac_small = np.bincount(hit, minlength=6) / hit.size, np.bincount(small["hit"], minlength=6) / small["hit"].size
    assert np.abs(frac - frac_small).max() < 5e-3, (frac, frac_small)
    assert int(ctr[0]) == size * size and int(ctr[4]) == size * size and int(ctr[6]) == 0
    steps_small = (small["counters"]["accepted"] + small["counters"]["rejected"]) / small["hit"].size
    assert abs((int(ctr[1]) + int(ctr[2])) / hit.size - steps_small) < 0.01 * steps_small
    b = sharded.trace_slab_torch(sc, opt, cam, size, size, size // 2, size // 2 + 64)
    torch.cuda.synchronize()
    assert torch.equal(b["rgb"], a["rgb"][:, (size // 2) * size:(size // 2 + 64) * size])
    j0, js, nr = sharded.row_assignment(size, 4, 1, "cyclic")
    share = {}
    sharded.trace_rows_torch(sc, opt, cam, size, size, j0, js, nr, out=share)
    torch.cuda.synchronize()
    assert torch.equal(share["rgb"].view(3, nr, size), a["rgb"].view(3, size, size)[:, j0::js, :])


def test_oracle_object_methods_known_answers():
    """(CPU) distance / min_distance / the colour rule of the oracle at hand-computed points: the reference's Plane and Sphere
    (src/RayTraceGR.jl:394-428), the inside-out sky, and the two user-object twins — zero ON the surface, the contract's signs, the
    colour rule's `omin / length(objs)` scale (:530) and its miss colour (:528)."""
    _, (torus, egg, _) = user_shapes()
    objs = [rt.Sphere((0, 0, 0, 0), (1, 0, 0, 0), -10), rt.Plane(-20), rt.Sphere((0, 4, 0, 0), (1, 0, 0, 0), 0.5), torus, egg]
    sc = rt.make_scene(rt.minkowski, objs, units=False)
    opt = rt.solver_defaults()
    x = np.array([[-3.0, 4.0, 1.0, 0.0],       # above the small sphere: |Δ|² − R² = 1 − 1/4
                  [-20.0, 0.0, 0.0, 0.0],      # ON the plane t = −20, inside the sky sphere: −(0 − 100) … sign(R) = −1
                  [0.0, 4.9, 0.0, 0.3],        # on the torus' surface: top of the tube at ϱ = R
                  [0.0, 4.0, 0.0, 0.0],        # the torus' centre hole: (0 − 0.9)² − 0.3² = 0.72
                  [0.0, 3.3, 1.5, -0.8],       # on the ellipsoid: one semi-axis b = 0.5 along y
                  [0.0, 3.3, 1.0, -0.8],       # the ellipsoid's centre: −1
                  [0.0, 4.0, 0.0, 0.7]])       # above the small sphere, near nothing: a miss
    r = O.eval_objects(sc, opt, x)
    assert np.allclose(r["d"][0, :3], [100.0 - 17.0, 17.0, 0.75], atol=1e-13)
    assert r["d"][1, 1] == 0.0 and np.isclose(r["d"][1, 0], 100.0) and r["hit"][1] == 2 and np.allclose(r["rgb"][1], np.array([0, 0.5, 0]) * 2 / 5)
    assert abs(r["d"][2, 3]) < 1e-15 and np.isclose(r["d"][3, 3], 0.72) and abs(r["d"][4, 4]) < 1e-15 and np.isclose(r["d"][5, 4], -1.0)
    assert r["hit"][2] == 4 and np.isclose(r["rgb"][2, 2], 0.5 * 4 / 5)                      # torus: blue = 1/2 x omin / nobj
    v = r["rgb"][2, 1] * 5 / 4                                                                 # poloidal angle π/2 at the top of the tube:
    assert min(v, 1.0 - v) < 1e-9                                                              # mod(6·(π/2)/π, 1) = mod(3, 1) — on the sawtooth's jump
    assert r["hit"][5] == 5 and r["hit"][3] == 3                                               # (the torus' hole holds the small sphere's centre)
    assert r["hit"][6] == 0 and np.allclose(r["rgb"][6], [1, 0, 0]) and r["dmin"][6] > 0.2     # nothing within the threshold: the miss colour (:528)
    assert np.allclose(r["dmin"], r["d"].min(axis=1))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-13), (np.float32, 2e-5)])
def test_object_methods_on_the_device_match_the_oracle_pointwise(lib, dtype, tol):
    """rtgr_eval_objects_*: distance of every object, min_distance and the colour rule at 4096 random points, built-in objects through
    the library's kernel and the user objects through their unit's, against the oracle — pointwise, not through a traced frame (the
    a6 / a7 rows of SURVEY §8 the way rtgr_eval_metric / rtgr_eval_geodesic serve a2–a5)."""
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-25, 2, (4096, 1)), rng.normal(size=(4096, 3)) * 2.5 + np.array([3.5, 0.5, 0.0])], axis=1)
    _, (torus, egg, torus2) = user_shapes()
    builtin = [rt.Sphere((0, 0, 0, 0), (1, 0, 0, 0), -10), rt.Plane(-20), rt.Sphere((0, 4, 0, 0), (1, 0, 0, 0), 0.5), rt.Disk(0.05, 2.0, 4.0)]
    opt = rt.solver_defaults(dtype, hit_threshold=0.05)
    for metric, objs in ((rt.kerr_schild, builtin), (rt.kerr_schild, builtin[:2] + [torus, egg, torus2]), (rt.KerrSchild(1.0, 0.8), [egg, builtin[0], torus])):
        got = rt.eval_objects(metric, objs, x, opt=opt, dtype=dtype)
        sco = rt.make_scene(metric, objs, units=False)
        ref = O.eval_objects(sco, opt, x, dtype=dtype)
        scale = 1.0 + np.abs(ref["d"].astype(float))
        assert (np.abs(got["d"].astype(float) - ref["d"]) / scale).max() <= tol
        assert (np.abs(got["dmin"].astype(float) - ref["dmin"]) / (1.0 + np.abs(ref["dmin"].astype(float)))).max() <= tol
        # the hit decision flips only where some distance sits within rounding of the threshold or of another object's; colours are
        # compared where the decisions agree (wrap-aware: a sawtooth may sit on either side of its jump)
        same = got["hit"] == ref["hit"]
        assert same.mean() >= 0.999 and same.sum() > 4000 and (got["hit"] > 0).sum() > 50
        dc = np.abs(got["rgb"][same].astype(float) - ref["rgb"][same])
        per = np.where(got["hit"][same] > 0, got["hit"][same] / len(objs), 1.0)[:, None]
        assert np.minimum(dc, np.abs(per - dc)).max() <= (1e-9 if dtype == np.float64 else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("which", [5, 6])
def test_example_scripts_with_user_objects_write_the_oracles_image(lib, tmp_path, monkeypatch, which):
    """examples/render.py 5 / 6 — what a user writes: new Object subtypes in `objs`, one family or two — end to end through make_canvas,
    trace_rays and the PNG writer; the 8-bit image equals the oracle's of the same scene (a pixel or two may sit on a rounding edge)."""
    import runpy
    from raytracegr_jl_amd.png import read_png
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, "argv", ["render.py", str(which), "64"])
    runpy.run_path(os.path.join(ROOT, "examples", "render.py"), run_name="__main__")
    img = read_png(str(tmp_path / "scenes" / f"sphere{which}.png"))
    _, objs, cam = rt.example2_scene()
    shapes = rt.UserObjects(user_objects.SHAPES)
    mine = [shapes(user_objects.TORUS, [4.0, 0.0, 0.0, 0.9, 0.3]), shapes(user_objects.ELLIPSOID, [3.3, 1.0, -0.8, 0.7, 0.5, 0.5])]
    if which == 6:
        mine.insert(1, rt.Sphere([0.0, 4.6, -0.9, 0.9], [1.0, 0.0, 0.0, 0.0], 0.35))
    ref = O.trace(rt.make_scene(rt.kerr_schild, objs[:2] + mine, units=False), rt.solver_defaults(), 64, 64, cam=rt.make_camera(**cam))
    gold = O.image_u8(ref["rgb"], 64, 64)
    assert img.shape == gold.shape == (64, 64, 3)
    off = (np.abs(img.astype(int) - gold.astype(int)) > 1).any(axis=2)
    assert int(off.sum()) <= 2, int(off.sum())
    assert (ref["hit"] >= 3).sum() > 400                      # the new objects fill a good part of the frame
