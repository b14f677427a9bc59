"""Shared scene builders for tests (inputs only; no reference code)."""
import numpy as np

from conftest import load_package

rt = load_package()


def example(n):
    metric, objs, cam = (rt.example1_scene if n == 1 else rt.example2_scene)()
    return rt.make_scene(metric, objs), rt.make_camera(**cam)


def user_shapes(reach=True, jit=False):
    """The two Object subtypes of examples/user_objects.py as a UserObjects family (one instance per (reach, jit): make_scene wants
    the user objects of a scene to share their family), and the two objects the *_shapes scenes put where example2 has its
    small sphere: a torus seen edge-on, an ellipsoid partly behind it and a second, smaller torus."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import user_objects as uo
    key = (bool(reach), bool(jit))
    if key not in _FAMILIES:
        _FAMILIES[key] = rt.UserObjects(uo.SHAPES_WITH_REACH if reach else uo.SHAPES, name="shapes" + ("+reach" if reach else ""), jit=jit)
    fam = _FAMILIES[key]
    # (the screen's rays fan out to x in [2.5, 5.5], z in [-1.5, 1.5] at y = 0: the three objects cover about a third of the frame)
    return fam, [fam(uo.TORUS, [4.0, 0.0, 0.0, 0.9, 0.3]), fam(uo.ELLIPSOID, [3.3, 1.0, -0.8, 0.7, 0.5, 0.5]),
                 fam(uo.TORUS, [4.7, 0.6, 0.95, 0.45, 0.2])]


_FAMILIES = {}


def many_objects(nobj, seed=7):
    """`objs::Vector{Object{T}}` has no length limit in the reference (src/RayTraceGR.jl:433-441, :483, :520-526): example2's sky
    sphere and far plane, then nobj - 2 seeded small spheres scattered around the hole (none contains example2's camera, none the
    horizon).  The list's ORDER carries the colour scale omin / length(objs) (:530) and first-smaller-wins (:520-526)."""
    rng = np.random.default_rng(seed)
    objs = [rt.Sphere((0, 0, 0, 0), (1, 0, 0, 0), -12.0), rt.Plane(-25.0)]
    cam = np.array([4.0, -2.0, 0.0])
    while len(objs) < nobj:
        d = rng.normal(size=3)
        c = d / np.linalg.norm(d) * rng.uniform(2.8, 8.5) * np.array([1.0, 1.0, 0.6])
        r = rng.uniform(0.15, 0.55)
        if np.linalg.norm(c - cam) < r + 0.3 or np.linalg.norm(c) < 2.5 + r:
            continue
        objs.append(rt.Sphere((0, *c), (1, 0, 0, 0), r))
    # the LAST object of the list stands in front of example2's camera (inside its fan of rays, also along straight lines): whatever
    # the length, the frame holds hits of the highest index — beyond the 16 inline slots for every list longer than that
    objs[-1] = rt.Sphere((0, 4.3, 0.5, 0.3), (1, 0, 0, 0), 0.2)
    return objs


def scene_variant(name, units=True, reach=True, jit=False):
    """BASELINE.json configs: 'ks_ref0' (example2 as written), 'ks_ref08', 'ks_true0', 'ks_true08', 'ks_true0998',
    'ks_true0998_disk' (config 5), 'mink' (example1).  '<variant>_shapes': the same with example2's small sphere replaced by two
    user-defined objects (user_shapes; units=False: a scene for the oracle — nothing is compiled or loaded)."""
    if "_many" in name:   # '<variant>_many<N>': N objects (many_objects) in place of example2's three, same camera
        base, n = name.split("_many")
        metric = {"ks_ref0": rt.kerr_schild, "ks_true08": rt.KerrSchild(1, 0.8), "ks_ref0_generic": rt.KerrSchild(1, 0.0, textbook=False, generic=True),
                  "mink": rt.minkowski}[base]
        return rt.make_scene(metric, many_objects(int(n)), units=units), rt.make_camera(**rt.example2_scene()[2])
    if name.endswith("_shapes"):
        base = name[:-len("_shapes")]
        metric, objs, cam = rt.example1_scene() if base == "mink" else rt.example2_scene()
        if base != "mink":
            metric = {"ks_ref0": rt.kerr_schild, "ks_ref08": rt.KerrSchild(1, 0.8, textbook=False), "ks_true0": rt.KerrSchild(1, 0.0),
                      "ks_true08": rt.KerrSchild(1, 0.8), "ks_ref0_generic": rt.KerrSchild(1, 0.0, textbook=False, generic=True)}[base]
        else:
            cam = rt.example2_scene()[2]   # (example2's camera: the shapes sit around (4, 0, 0))
        return rt.make_scene(metric, objs[:2] + user_shapes(reach, jit)[1], units=units), rt.make_camera(**cam)
    _, objs, cam = rt.example2_scene()
    m = {"ks_ref0": rt.kerr_schild, "ks_ref08": rt.KerrSchild(1, 0.8, textbook=False),
         "ks_true0": rt.KerrSchild(1, 0.0), "ks_true08": rt.KerrSchild(1, 0.8),
         "ks_true0998": rt.KerrSchild(1, 0.998), "ks_true0998_disk": rt.KerrSchild(1, 0.998)}.get(name)
    if name == "mink":
        return example(1)
    if name == "ks_true0998_disk":
        objs = objs[:2] + [rt.Disk(0.05, 2.0, 4.0)]  # camera (cylindrical radius 4.5) stays outside the disk
    return rt.make_scene(m, objs), rt.make_camera(**cam)


def circular_channels(hit, sc=None):
    """[3, n] bool: which RGB channels of a pixel carry a SAWTOOTH mod(·, 1)·omin/nobj and are therefore compared circularly.
    A sphere hit puts it on R and G (objcolor, src/RayTraceGR.jl:420-428: B is the constant 1), the thin disk (no reference
    counterpart; rtgr_integrator.hpp colour_pixel) on G and B (R is the constant 1); a plane's colour is constant.  Without a
    scene every hit is taken for a sphere.  (the helper used to wrap B for every object, which scored a real B error
    close to a multiple of omin/nobj on a sphere as ~0.)"""
    n = hit.shape[0]
    circ = np.zeros((3, n), bool)
    kinds = np.full(n, rt._abi.SPHERE)
    if sc is not None:
        table = np.array([0] + [sc.object(o).kind for o in range(sc.nobj)])
        kinds = table[np.minimum(hit, sc.nobj)]
    sph, dsk = (kinds == rt._abi.SPHERE) & (hit > 0), (kinds == rt._abi.DISK) & (hit > 0)
    circ[0] = sph
    circ[1] = sph | dsk
    circ[2] = dsk
    if sc is not None:   # the user objects of examples/user_objects.py: torus (type 0) R and G, ellipsoid (type 1) R and B
        types = np.array([0] + [sc.object(o).type for o in range(sc.nobj)])[np.minimum(hit, sc.nobj)]
        usr = (kinds == rt._abi.USER_OBJECT) & (hit > 0)
        circ[0] |= usr
        circ[1] |= usr & (types == 0)
        circ[2] |= usr & (types == 1)
    return circ


def wrap_aware_rgb_err(a, b, hit, nobj=3, sc=None, per_pixel=False):
    """L∞ distance between RGB planes a, b [3, n], evaluated modulo the sawtooth of objcolor (src/RayTraceGR.jl:427) on the
    channels that carry one (circular_channels): there the circular distance has period omin/nobj; every other channel is
    compared plainly.  Pass the scene when it may hold a disk."""
    if sc is not None:
        nobj = sc.nobj
    d = np.abs(a - b)
    per = (hit.astype(np.float64) / max(nobj, 1))[None, :]
    per = np.where(per > 0, per, 1.0)
    dc = np.where(circular_channels(hit, sc), np.minimum(d, np.abs(per - d)), d)
    return dc.max(axis=0) if per_pixel else dc.max(initial=0.0)
