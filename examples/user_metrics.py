"""User-metric sources (api.UserMetric): metrics written the way a user of the reference would write a new Julia
metric function next to `kerr_schild` (src/RayTraceGR.jl:274-294), as C++ templates over the scalar type."""

# textbook Kerr–Schild, the same function as the built-in KerrSchild(M, a, textbook=True): η + f k⊗k
KERR_SCHILD = r'''
template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]) {
    const S X = x[1], Y = x[2], Z = x[3];
    const double a2 = a * a;
    const S rho2 = X * X + Y * Y + Z * Z;
    const S q = rho2 - a2;
    const S r2 = 0.5 * (q + msqrt(q * q + (4.0 * a2) * (Z * Z)));
    const S r = msqrt(r2);
    const S f = ((2.0 * M) * (r2 * r)) / (r2 * r2 + a2 * (Z * Z));
    const S den = r2 + a2;
    S k[4];
    k[0] = mconst<S>(1.0);
    k[1] = (r * X + a * Y) / den;
    k[2] = (r * Y - a * X) / den;
    k[3] = Z / r;
    for (int p = 0; p < 4; p++)
        for (int c = 0; c < 4; c++) {
            g[p][c] = f * k[p] * k[c];
            if (p == c) g[p][c] = g[p][c] + (p == 0 ? -1.0 : 1.0);
        }
}
'''

# The same metric given in KERR–SCHILD FORM (rtgr_user_ks): only f and k — what is different between one Kerr–Schild metric
# and another.  The library derives g = eta + f k k for the camera and the evaluation hooks, and the integrate kernels
# differentiate these four scalars (instead of ten metric entries) and use the closed contraction of the built-ins.
# Contract: k_t = 1, k null with respect to eta (|k| = 1), no t-dependence.
KERR_SCHILD_KS = r'''
template <class S> __device__ void rtgr_user_ks(const S x[4], double M, double a, S& f, S k[3]) {
    const S X = x[1], Y = x[2], Z = x[3];
    const double a2 = a * a;
    const S ZZ = Z * Z;
    const S q = X * X + Y * Y + ZZ - a2;
    const S r2 = 0.5 * (q + msqrt(q * q + (4.0 * a2) * ZZ));
    const S r = msqrt(r2);
    f = ((2.0 * M) * (r2 * r)) / (r2 * r2 + a2 * ZZ);
    const S den = r2 + a2;
    k[0] = (r * X + a * Y) / den;
    k[1] = (r * Y - a * X) / den;
    k[2] = Z / r;
}
'''

# Schwarzschild in isotropic coordinates — NOT of Kerr–Schild form, so no built-in covers it:
#   ds² = −((1−m)/(1+m))² dt² + (1+m)⁴ (dx²+dy²+dz²),  m = M / (2ρ)
SCHWARZSCHILD_ISOTROPIC = r'''
template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]) {
    const S rho = msqrt(x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    const S m = (0.5 * M) / rho;
    const S lapse = (1.0 - m) / (1.0 + m);
    const S psi2 = (1.0 + m) * (1.0 + m);
    for (int p = 0; p < 4; p++)
        for (int c = 0; c < 4; c++) g[p][c] = mconst<S>(0.0);
    g[0][0] = -(lapse * lapse);
    g[1][1] = g[2][2] = g[3][3] = psi2 * psi2;
}
'''

# A TIME-DEPENDENT metric: isotropic Schwarzschild whose spatial part expands, g_ij = e^{2Ht} psi^4 delta_ij, H = the metric's
# `a` parameter.  Not a solution of anything — a checker (oracle/rtgr_oracle.cpp expanding_isotropic): a metric that depends on
# t must be traced with d_t g != 0 evaluated at the ray's own t, as the reference does (src/RayTraceGR.jl:302-313, :358-363);
# UserMetric(..., stationary=False) (the default) carries the stage time through the integrate kernels for it.
EXPANDING_ISOTROPIC = r'''
template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]) {
    const S rho = msqrt(x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    const S m = (0.5 * M) / rho;
    const S lapse = (1.0 - m) / (1.0 + m);
    const S psi2 = (1.0 + m) * (1.0 + m);
    const S s = mexp(a * x[0]);
    for (int p = 0; p < 4; p++)
        for (int c = 0; c < 4; c++) g[p][c] = mconst<S>(0.0);
    g[0][0] = -(lapse * lapse);
    g[1][1] = g[2][2] = g[3][3] = (s * s) * (psi2 * psi2);
}
'''

# A smooth, static, non-diagonal perturbation of flat space written with EVERY elementary function the reference's Dual
# carries (src/RayTraceGR.jl:132-196) — not physics, a checker: the oracle has the same function (oracle/rtgr_oracle.cpp
# helper_zoo) and tests compare g, dg, Christoffels and the RHS point by point.
HELPER_ZOO = r'''
template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]) {
    const S X = x[1], Y = x[2], Z = x[3];
    const S rho = msqrt(X * X + Y * Y + Z * Z);
    const S cth = Z / rho;
    const S th = macos(cth), ph = matan2(Y, X), lat = masin(cth);
    for (int p = 0; p < 4; p++)
        for (int c = 0; c < 4; c++) g[p][c] = mconst<S>(0.0);
    g[0][0] = mconst<S>(-1.0) - (0.1 * M) * mpow(rho, -1.5);
    g[1][1] = mconst<S>(1.0) + 0.05 * mabs(lat) + 0.02 * mexp(-rho);
    g[2][2] = mconst<S>(1.0) + 0.05 * mcbrt(1.0 + rho) / rho;
    g[3][3] = mconst<S>(1.0) + 0.03 * matan(rho) * mlog(2.0 + rho) / rho;
    g[1][2] = g[2][1] = 0.01 * msin(ph) * mcos(th);
    g[0][3] = g[3][0] = (0.02 * M) * cth / rho;
}
'''

# Kerr in Boyer–Lindquist coordinates (t, r, θ, φ), drawn on the tracer's Cartesian-like coordinates by the plain
# spherical map x = r sinθ cosφ, y = r sinθ sinφ, z = r cosθ — the metric a user reaches for first, and one that needs the
# inverse trigonometric helpers (macos, matan2) the reference's Dual carries (src/RayTraceGR.jl:154-169):
#   Σ = r² + a² cos²θ,  Δ = r² − 2Mr + a²
#   ds² = −(1 − 2Mr/Σ) dt² − (4Mar sin²θ/Σ) dt dφ + (Σ/Δ) dr² + Σ dθ² + (r² + a² + 2Ma²r sin²θ/Σ) sin²θ dφ²
# g_ab = Σ_μν (∂q^μ/∂x^a)(∂q^ν/∂x^b) g_μν with the rows dr = (sθcφ, sθsφ, cθ), dθ = (cθcφ, cθsφ, −sθ)/r, dφ = (−sφ, cφ, 0)/(r sθ).
# Stationary; singular on the axis of the spherical map and at the horizon Δ = 0 (where captured rays hover, as in isotropic
# Schwarzschild).  No oracle twin: checked against an independent sympy + DOP853 solution (tests/truth.py 'kerr_bl').
# Cost (1024², a = 0.8): 1.4e9 step attempts/s against 9.3e9 for KERR_SCHILD above — the four f64 sin/cos and the acos /
# atan2 of dual numbers are library calls of ~100 instructions each, and Boyer–Lindquist rays need 770 steps (captured rays
# crawl towards Δ = 0).  Written for clarity; cosθ = z/r, sinθ = ϖ/r, cosφ = x/ϖ, sinφ = y/ϖ is the fast way to type it.
KERR_BOYER_LINDQUIST = r"""
template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]) {
    const S X = x[1], Y = x[2], Z = x[3];
    const S r = msqrt(X * X + Y * Y + Z * Z);
    const S th = macos(Z / r), ph = matan2(Y, X);
    const S st = msin(th), ct = mcos(th), sp = msin(ph), cp = mcos(ph);
    const double a2 = a * a;
    const S r2 = r * r, s2 = st * st;
    const S Sig = r2 + a2 * (ct * ct);
    const S Del = r2 - (2.0 * M) * r + a2;
    const S w = ((2.0 * M) * r) / Sig;                       // 2Mr/Σ
    const S gtt = w - 1.0, gtp = -(a * w) * s2, grr = Sig / Del, gthth = Sig;
    const S gpp = (r2 + a2 + (a2 * w) * s2) * s2;
    const S zero = mconst<S>(0.0);
    const S ir = 1.0 / r, irs = ir / st;
    const S dr[4] = {zero, st * cp, st * sp, ct};
    const S dth[4] = {zero, ct * cp * ir, ct * sp * ir, -(st * ir)};
    const S dph[4] = {zero, -(sp * irs), cp * irs, zero};
    for (int p = 1; p < 4; p++)
        for (int c = 1; c < 4; c++) g[p][c] = grr * dr[p] * dr[c] + gthth * dth[p] * dth[c] + gpp * dph[p] * dph[c];
    g[0][0] = gtt;
    for (int p = 1; p < 4; p++) g[0][p] = g[p][0] = gtp * dph[p];
}
"""
