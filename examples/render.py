#!/usr/bin/env python3
"""Drop-in demo: the reference's example1() / example2() (src/RayTraceGR.jl:542-612) through the HIP library.

    python examples/render.py 2 [ni nj]     # writes scenes/sphere2.png (example2: black hole), default 200 x 200
    python examples/render.py 1             # writes scenes/sphere.png  (example1: flat space)
    python examples/render.py 3             # writes scenes/sphere3.png (example2's scene around a Schwarzschild hole
                                            #   in isotropic coordinates: a metric compiled at run time, UserMetric)
    python examples/render.py 4             # writes scenes/sphere4.png (the same scene around a Kerr hole, a = 0.8, in
                                            #   Boyer–Lindquist coordinates — user source with macos / matan2)
    python examples/render.py 5             # writes scenes/sphere5.png (example2 with NEW Object subtypes — a torus and an
                                            #   ellipsoid given as device source, UserObjects — where the small sphere was)
    python examples/render.py 6             # writes scenes/sphere6.png (objects of TWO separately written sources in one scene:
                                            #   that torus and ellipsoid, and the reference's Sphere written as device source)

The code below is what a user of RayTraceGR.jl writes, with `RayTraceGR.` replaced by the host mirror `rt.`.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

rt = load_package()


def main():
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ni = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    nj = int(sys.argv[3]) if len(sys.argv) > 3 else ni
    metric = rt.minkowski if which == 1 else rt.kerr_schild
    if which in (3, 4):  # a metric function of the user's own (the reference: any Julia callable, src/RayTraceGR.jl:302-309)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import user_metrics
        metric = (rt.UserMetric(user_metrics.SCHWARZSCHILD_ISOTROPIC, M=1.0, name="schwarzschild_isotropic") if which == 3 else
                  rt.UserMetric(user_metrics.KERR_BOYER_LINDQUIST, M=1.0, a=0.8, stationary=True, name="kerr_boyer_lindquist"))
    caelum = rt.Sphere((0, 0, 0, 0), (1, 0, 0, 0), -10)                       # background sky, inside-out
    frustum = rt.Plane(-20)                                                    # cut-off plane in the past
    sphere = rt.Sphere((0, 0 if which == 1 else 4, 0, 0), (1, 0, 0, 0), 0.5)   # the visible sphere
    objs = [caelum, frustum, sphere]
    if which == 5:   # new subtypes of the reference's open `abstract type Object{T}` (distance / objcolor, src/RayTraceGR.jl:374-389)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import user_objects
        shapes = rt.UserObjects(user_objects.SHAPES_WITH_REACH, name="torus + ellipsoid")
        objs = [caelum, frustum, shapes(user_objects.TORUS, [4.0, 0.0, 0.0, 0.9, 0.3]), shapes(user_objects.ELLIPSOID, [3.3, 1.0, -0.8, 0.7, 0.5, 0.5])]
    if which == 6:   # … of two families at once: make_scene joins the sources into one unit (each family says how many types it defines)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import user_objects
        shapes = rt.UserObjects(user_objects.SHAPES_WITH_REACH, name="torus + ellipsoid", ntypes=2)
        balls = rt.UserObjects(user_objects.SPHERE_AS_USER_OBJECT, name="sphere", ntypes=1)
        objs = [caelum, frustum, shapes(user_objects.TORUS, [4.0, 0.0, 0.0, 0.9, 0.3]), balls(0, [0.0, 4.6, -0.9, 0.9, 1.0, 0.0, 0.0, 0.0, 0.35]),
                shapes(user_objects.ELLIPSOID, [3.3, 1.0, -0.8, 0.7, 0.5, 0.5])]
    pos = (0, 0 if which == 1 else 4, -2, 0)
    if which in (5, 6):   # the sources bring a reach bound (it lets the FAR pass skip scans): once per scene, hold it against the single FULL pass
        rt.check_scene(metric, objs, dict(pos=pos, widthx=(0, 1, 0, 0), widthy=(0, 0, 0, 1), normal=(0, 0, 1, 0)))
    canvas = rt.make_canvas(metric, pos, (0, 1, 0, 0), (0, 0, 0, 1), (0, 0, 1, 0), ni, nj)
    canvas, info = rt.trace_rays(metric, objs, canvas, return_info=True)
    from raytracegr_jl_amd.png import write_png
    os.makedirs(rt.api.outdir, exist_ok=True)
    file = os.path.join(rt.api.outdir, {1: "sphere.png", 2: "sphere2.png", 3: "sphere3.png", 4: "sphere4.png", 5: "sphere5.png", 6: "sphere6.png"}[which])
    write_png(file, canvas.image_u8())
    print(f'Output file is "{file}"  ({info["rays"]} rays, {info["accepted"] + info["rejected"]} RK step attempts)')


if __name__ == "__main__":
    main()
