"""Shared scene builders for tests (inputs only; no reference code)."""
import numpy as np

from conftest import load_package

rt = load_package()


def example(n):
    metric, objs, cam = (rt.example1_scene if n == 1 else rt.example2_scene)()
    return rt.make_scene(metric, objs), rt.make_camera(**cam)


def scene_variant(name):
    """BASELINE.json configs: 'ks_ref0' (example2 as written), 'ks_ref08', 'ks_true0', 'ks_true08', 'ks_true0998',
    'ks_true0998_disk' (config 5), 'mink' (example1)."""
    _, objs, cam = rt.example2_scene()
    m = {"ks_ref0": rt.kerr_schild, "ks_ref08": rt.KerrSchild(1, 0.8, textbook=False),
         "ks_true0": rt.KerrSchild(1, 0.0), "ks_true08": rt.KerrSchild(1, 0.8),
         "ks_true0998": rt.KerrSchild(1, 0.998), "ks_true0998_disk": rt.KerrSchild(1, 0.998)}.get(name)
    if name == "mink":
        return example(1)
    if name == "ks_true0998_disk":
        objs = objs[:2] + [rt.Disk(0.05, 2.0, 4.0)]  # camera (cylindrical radius 4.5) stays outside the disk
    return rt.make_scene(m, objs), rt.make_camera(**cam)


def wrap_aware_rgb_err(a, b, hit, nobj=3):
    """L∞ distance between RGB planes a, b [3, n], evaluated modulo the sawtooth of objcolor (src/RayTraceGR.jl:427):
    channels R,G of a sphere hit are mod(·,1)*omin/nobj, so the circular distance has period omin/nobj.  The thin disk (no
    reference counterpart) puts its sawtooths on G and B (rtgr_integrator.hpp colour_pixel), so all three channels are
    compared circularly; on a channel that is constant for the object hit the circular distance IS the distance."""
    d = np.abs(a - b)
    per = (hit.astype(np.float64) / nobj)[None, :]
    per = np.where(per > 0, per, 1.0)
    dc = np.minimum(d, np.abs(per - d))
    return dc.max(initial=0.0)
