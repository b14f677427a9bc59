"""New `Object{T}` subtypes as device source (api.UserObjects; include/rtgr.h "user objects").

The reference's objects are an OPEN abstract type: anything with `distance(obj, pos)` — zero on the surface, positive outside,
negative inside — and `objcolor(obj, pos)` is traced (src/RayTraceGR.jl:374-389, :433-441, :518-530).  SHAPES defines two such
types the reference does not have; oracle/rtgr_oracle.cpp carries their CPU twins (RTGR_USER_OBJECT, same type tags).

    shapes = rt.UserObjects(SHAPES)
    torus = shapes(TORUS, [4, 0, 0,  0.9, 0.3])            # centre (x, y, z), major radius, minor radius — axis along z
    egg   = shapes(ELLIPSOID, [3.3, 1, -0.8,  0.7, 0.5, 0.5])   # centre, semi-axes
"""
TORUS, ELLIPSOID = 0, 1

SHAPES = r'''
template <class S> __device__ S rtgr_user_distance(unsigned type, const S x[4], const S p[9]) {
    const S X = x[1] - p[0], Y = x[2] - p[1], Z = x[3] - p[2];
    if (type == 0u) {                                        // torus around the z axis: (sqrt(X² + Y²) − R)² + Z² − r²
        const S w = msqrt(X * X + Y * Y) - p[3];
        return w * w + Z * Z - p[4] * p[4];
    }
    const S xs = X / p[3], ys = Y / p[4], zs = Z / p[5];     // ellipsoid: (X/a)² + (Y/b)² + (Z/c)² − 1
    return xs * xs + ys * ys + zs * zs - S(1);
}
template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]) {
    const S pi = S(3.14159265358979323846264338327950288);
    const S X = x[1] - p[0], Y = x[2] - p[1], Z = x[3] - p[2];
    if (type == 0u) {                                        // toroidal and poloidal angle
        const S w = msqrt(X * X + Y * Y) - p[3];
        rgb[0] = mod1<S>(S(6) * matan2(Y, X) / pi);
        rgb[1] = mod1<S>(S(6) * matan2(Z, w) / pi);
        rgb[2] = S(0.5);
        return;
    }
    const S xs = X / p[3], ys = Y / p[4], zs = Z / p[5];     // the sphere's rule (:420-428) in the scaled coordinates
    const S r = msqrt(xs * xs + ys * ys + zs * zs);
    rgb[0] = mod1<S>(S(12) * macos(zs / r) / pi);
    rgb[1] = S(0.5);
    rgb[2] = mod1<S>(S(12) * matan2(ys, xs) / pi);
}
'''

# … and the bound that lets the FAR pass skip a step's scan: how far the distance can move inside the box |x'_q − x_q| <= dl[q]
REACH = r'''
template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]) {
    const S X = x[1] - p[0], Y = x[2] - p[1], Z = x[3] - p[2];
    if (type == 0u) {   // |Δ(ϱ − R)| <= |Δ(X, Y)| <= sqrt(dl_x² + dl_y²) =: d;  |Δ w²| <= d (2|w| + d);  |Δ Z²| <= dl_z (2|Z| + dl_z)
        const S w = msqrt(X * X + Y * Y) - p[3];
        const S d = msqrt(dl[1] * dl[1] + dl[2] * dl[2]);
        return d * (S(2) * mabs(w) + d) + dl[3] * (S(2) * mabs(Z) + dl[3]);
    }
    const S ex = dl[1] / p[3], ey = dl[2] / p[4], ez = dl[3] / p[5];
    return ex * (S(2) * mabs(X / p[3]) + ex) + ey * (S(2) * mabs(Y / p[4]) + ey) + ez * (S(2) * mabs(Z / p[5]) + ez);
}
'''

SHAPES_WITH_REACH = SHAPES + REACH

# … and a sample object per type for the library's load-time probe (include/rtgr.h "rtgr_user_sample"): about 0.5 across, around
# (x, y, z) = (4, 0, 0) — where example2's small sphere stands.  With it a unit whose reach bound disagrees with its distance on
# the samples does not even load; without it the automatic scene check of the first trace is what catches such a bound.
SAMPLE = r'''
template <class S> __device__ bool rtgr_user_sample(unsigned type, S p[9]) {
    if (type == 0u) { p[0] = S(4.0); p[1] = S(0.0); p[2] = S(0.0); p[3] = S(0.32); p[4] = S(0.11); return true; }      // a small torus
    if (type == 1u) { p[0] = S(3.9); p[1] = S(0.3); p[2] = S(0.35); p[3] = S(0.22); p[4] = S(0.15); p[5] = S(0.18); return true; }   // an egg beside it
    return false;
}
'''
SHAPES_WITH_REACH_AND_SAMPLES = SHAPES + REACH + SAMPLE

# The reference's own Sphere (src/RayTraceGR.jl:409-428) typed as a USER object — same distance, same colour rule, and the reach
# bound the library uses for its built-in sphere: what the generic dispatch of a user object costs against the built-in one is
# the difference between two frames that are otherwise the same (bench.py: variants.user_sphere_*).  p = pos (4), vel (4), radius.
SPHERE_AS_USER_OBJECT = r'''
template <class S> __device__ S rtgr_user_distance(unsigned type, const S x[4], const S p[9]) {
    const S dx = x[1] - p[1], dy = x[2] - p[2], dz = x[3] - p[3];
    const S d = rfma(dx, dx, rfma(dy, dy, rfma(dz, dz, -p[8] * p[8])));           // (the built-in's own order of fused operations)
    return p[8] < S(0) ? -d : d;                                                  // sign(R) * (|x - c|² - R²)   :415-419
}
template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]) {
    const S pi = S(3.14159265358979323846264338327950288);
    const S dx = x[1] - p[1], dy = x[2] - p[2], dz = x[3] - p[3], r = msqrt(dx * dx + dy * dy + dz * dz);
    rgb[0] = mod1<S>(S(12) * macos(dz / r) / pi);                                  // :420-428
    rgb[1] = mod1<S>(S(12) * matan2(dy, dx) / pi);
    rgb[2] = S(1);
}
template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]) {
    const S dx = x[1] - p[1], dy = x[2] - p[2], dz = x[3] - p[3];
    return dl[1] * (S(2) * mabs(dx) + dl[1]) + dl[2] * (S(2) * mabs(dy) + dl[2]) + dl[3] * (S(2) * mabs(dz) + dl[3]);
}
'''
