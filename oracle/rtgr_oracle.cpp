// rtgr_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A CPU restatement of the algorithm of eschnett/RayTraceGR.jl's per-pixel hot path, written to be checked
// against the reference's committed outputs (tests/golden/sphere.png, sphere2.png) and then used as the judge of
// the HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product (raytracegr.jl_amd/, librtgr_hip.so) never includes, links or calls anything in this directory.
//
// PARITY PIN: the reference's test suite holds no vector for trace_rays (test/runtests.jl never reaches it), so
// this oracle is pinned by the reference's committed output images: tests/test_oracle_golden.py requires
// sphere2.png 40000/40000 8-bit-exact and sphere.png exact outside the documented silhouette ring
// (SURVEY.md §0.5, §4.3), plus the hand-derived known answers of SURVEY.md §4.2.
//
// Each function cites the reference lines it follows (paths relative to /root/reference).  The integrator,
// step controller and event finder are NOT in /root/reference: they are OrdinaryDiffEq 5.38.3 / DiffEqBase 6.35.2 /
// Roots 1.0.1 / StaticArrays 0.12.3 as pinned by Manifest.toml; their published algorithms are restated from
// SURVEY.md App. A/B and anchored on the call site src/RayTraceGR.jl:488-511.
//
// The struct layouts come from include/rtgr.h so one ctypes description serves oracle and product alike.

#include "../include/rtgr.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <type_traits>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int D = 4;  // src/RayTraceGR.jl:253-254

// ---------------------------------------------------------------------------------------------------------------
// Dual{T,DT} with DT = SVector{4,T}                                                  src/RayTraceGR.jl:11-14
// Only the operations kerr_schild reaches are restated, each exactly as written.
// ---------------------------------------------------------------------------------------------------------------
template <class T>
struct Dual {
    T val;
    T eps[D];
};

template <class T>
inline Dual<T> mk(T v) {  // Dual{T,DT}(val) = Dual(val, zeros(DT))                             :16-21
    return Dual<T>{v, {T(0), T(0), T(0), T(0)}};
}
template <class T>
inline Dual<T> operator+(const Dual<T>& x, const Dual<T>& y) {  // :59-61
    Dual<T> r;
    r.val = x.val + y.val;
    for (int i = 0; i < D; i++) r.eps[i] = x.eps[i] + y.eps[i];
    return r;
}
template <class T>
inline Dual<T> operator+(const Dual<T>& x, T a) {  // :62-64, :68-70
    Dual<T> r = x;
    r.val = x.val + a;
    return r;
}
template <class T>
inline Dual<T> operator-(const Dual<T>& x, const Dual<T>& y) {  // :75-77
    Dual<T> r;
    r.val = x.val - y.val;
    for (int i = 0; i < D; i++) r.eps[i] = x.eps[i] - y.eps[i];
    return r;
}
template <class T>
inline Dual<T> operator-(const Dual<T>& x, T a) {  // :78-80, :84-86
    Dual<T> r = x;
    r.val = x.val - a;
    return r;
}
template <class T>
inline Dual<T> operator*(const Dual<T>& x, const Dual<T>& y) {  // :91-93  eps = x.eps.*y.val + x.val.*y.eps
    Dual<T> r;
    r.val = x.val * y.val;
    for (int i = 0; i < D; i++) r.eps[i] = x.eps[i] * y.val + x.val * y.eps[i];
    return r;
}
template <class T>
inline Dual<T> operator*(const Dual<T>& x, T a) {  // :94-96, :100-102
    Dual<T> r;
    r.val = x.val * a;
    for (int i = 0; i < D; i++) r.eps[i] = x.eps[i] * a;
    return r;
}
template <class T>
inline Dual<T> operator*(T a, const Dual<T>& x) {  // :97-99, :103-105
    Dual<T> r;
    r.val = a * x.val;
    for (int i = 0; i < D; i++) r.eps[i] = a * x.eps[i];
    return r;
}
template <class T>
inline Dual<T> operator/(const Dual<T>& x, const Dual<T>& y) {  // :112-114 (x.eps*y.val - x.val*y.eps)/y.val^2
    Dual<T> r;
    r.val = x.val / y.val;
    T y2 = y.val * y.val;
    for (int i = 0; i < D; i++) r.eps[i] = (x.eps[i] * y.val - x.val * y.eps[i]) / y2;
    return r;
}
template <class T>
inline Dual<T> operator/(const Dual<T>& x, T a) {  // :115-120
    Dual<T> r;
    r.val = x.val / a;
    for (int i = 0; i < D; i++) r.eps[i] = x.eps[i] / a;
    return r;
}
template <class T>
inline Dual<T> pow2(const Dual<T>& x) { return x * x; }  // literal_pow Val{2}  :134
template <class T>
inline Dual<T> pow3(const Dual<T>& x) { return x * x * x; }  // Val{3}          :135
template <class T>
inline Dual<T> pow4(const Dual<T>& x) { return pow2(pow2(x)); }  // Val{4}      :136
template <class T>
inline Dual<T> sqrt(const Dual<T>& x) {  // :193-196   r = sqrt(val); eps = 1/(2r) .* eps
    Dual<T> r;
    r.val = std::sqrt(x.val);
    T c = T(1) / (T(2) * r.val);
    for (int i = 0; i < D; i++) r.eps[i] = c * x.eps[i];
    return r;
}
// The remaining elementary functions of the reference's Dual (abs :150, acos asin atan :154-163, atan(y,x) :165-169,
// cbrt :171, ^ :138-148): value and TEXTBOOK derivative.  The hot path (minkowski / kerr_schild) reaches none of them;
// they are the checker of the product's user-metric helpers (mabs macos masin matan matan2 mcbrt mpow).  Divergence from
// the reference as written, on purpose and documented: its atan(y,x) drops a 1/ρ² on the first term
// (:165-169); that is a bug off the reference's hot path, not behaviour to be faithful to.
template <class T>
inline Dual<T> chain(const Dual<T>& x, T f, T df) {
    Dual<T> r;
    r.val = f;
    for (int i = 0; i < D; i++) r.eps[i] = df * x.eps[i];
    return r;
}
template <class T> inline Dual<T> abs(const Dual<T>& x) { return chain(x, std::abs(x.val), x.val < T(0) ? T(-1) : T(1)); }          // :150
template <class T> inline Dual<T> acos(const Dual<T>& x) { return chain(x, std::acos(x.val), T(-1) / std::sqrt(T(1) - x.val * x.val)); }  // :154
template <class T> inline Dual<T> asin(const Dual<T>& x) { return chain(x, std::asin(x.val), T(1) / std::sqrt(T(1) - x.val * x.val)); }   // :157
template <class T> inline Dual<T> atan(const Dual<T>& x) { return chain(x, std::atan(x.val), T(1) / (T(1) + x.val * x.val)); }            // :160
template <class T> inline Dual<T> atan2(const Dual<T>& y, const Dual<T>& x) {                                                       // :165-169
    Dual<T> r;
    r.val = std::atan2(y.val, x.val);
    T ir2 = T(1) / (x.val * x.val + y.val * y.val);
    for (int i = 0; i < D; i++) r.eps[i] = (x.val * y.eps[i] - y.val * x.eps[i]) * ir2;
    return r;
}
template <class T> inline Dual<T> cbrt(const Dual<T>& x) { T c = std::cbrt(x.val); return chain(x, c, T(1) / (T(3) * c * c)); }     // :171
template <class T> inline Dual<T> powr(const Dual<T>& x, T p) { T f = std::pow(x.val, p); return chain(x, f, p * f / x.val); }      // :138-148
template <class T> inline Dual<T> sin(const Dual<T>& x) { return chain(x, std::sin(x.val), std::cos(x.val)); }                      // :190
template <class T> inline Dual<T> cos(const Dual<T>& x) { return chain(x, std::cos(x.val), -std::sin(x.val)); }                     // :176
template <class T> inline Dual<T> exp(const Dual<T>& x) { T f = std::exp(x.val); return chain(x, f, f); }                           // :182
template <class T> inline Dual<T> log(const Dual<T>& x) { return chain(x, std::log(x.val), T(1) / x.val); }                         // :186
using std::abs; using std::acos; using std::asin; using std::atan; using std::atan2; using std::cbrt; using std::sin; using std::cos;
using std::exp; using std::log;
inline double powr(double x, double p) { return std::pow(x, p); }
inline float powr(float x, float p) { return std::pow(x, p); }
inline long double powr(long double x, long double p) { return std::pow(x, p); }
// plain-scalar twins so the metric below is one template over S = T or Dual<T>
inline double pow2(double x) { return x * x; }
inline double pow3(double x) { return x * x * x; }
inline double pow4(double x) { return (x * x) * (x * x); }
inline float pow2(float x) { return x * x; }
inline float pow3(float x) { return x * x * x; }
inline float pow4(float x) { return (x * x) * (x * x); }
using std::sqrt;

template <class T>
inline T valof(const T& x) { return x; }
template <class T>
inline T valof(const Dual<T>& x) { return x.val; }

// ---------------------------------------------------------------------------------------------------------------
// Metrics
// ---------------------------------------------------------------------------------------------------------------
template <class T, class S>
inline void eta(S g[D][D]) {  // @SMatrix T[a==b ? (a==1 ? -1 : 1) : 0 ...]            :263, :282
    for (int a = 0; a < D; a++)
        for (int b = 0; b < D; b++) {
            T v = (a == b) ? (a == 0 ? T(-1) : T(1)) : T(0);
            if constexpr (std::is_same<S, T>::value) g[a][b] = v; else g[a][b] = mk<T>(v);
        }
}

// minkowski(x)                                                                     src/RayTraceGR.jl:262-264
template <class T, class S>
inline void minkowski(const S[D], S g[D][D]) { eta<T, S>(g); }

// kerr_schild(xx) AS WRITTEN (variant 1) or with the textbook radius (variant 2)   src/RayTraceGR.jl:274-294
// The reference has the locals M = 1, a = 0 (:275-276); they are parameters here.
template <class T, class S>
inline void kerr_schild(const S xx[D], T M, T a, int variant, S g[D][D]) {
    const S &x = xx[1], &y = xx[2], &z = xx[3];                                   // t,x,y,z = xx          :278
    S e[D][D];
    eta<T, S>(e);                                                                   // η                     :282
    S rho = sqrt(pow2(x) + pow2(y) + pow2(z));                                      // ρ                     :283
    T a2 = a * a;
    S r;
    if (variant == RTGR_KS_REF) {
        // r = sqrt(ρ^2 - a^2)/2 + sqrt(a^2*z^2 + ((ρ^2 - a^2)/2)^2)                                        :284
        r = sqrt(pow2(rho) - a2) / T(2) + sqrt(a2 * pow2(z) + pow2((pow2(rho) - a2) / T(2)));
    } else {
        // textbook: r^2 = (q + sqrt(q^2 + 4 a^2 z^2))/2 with q = ρ^2 - a^2   (no reference counterpart)
        S q = pow2(rho) - a2;
        r = sqrt((q + sqrt(pow2(q) + (T(4) * a2) * pow2(z))) / T(2));
    }
    S f = (T(2) * M) * pow3(r) / (pow4(r) + a2 * pow2(z));                          // f                     :285
    S k[D];                                                                         // k                     :286-289
    if constexpr (std::is_same<S, T>::value) k[0] = T(1); else k[0] = mk<T>(T(1));
    k[1] = (r * x + a * y) / (pow2(r) + a2);
    k[2] = (r * y - a * x) / (pow2(r) + a2);
    k[3] = z / r;
    for (int p = 0; p < D; p++)
        for (int q = 0; q < D; q++) g[p][q] = e[p][q] + f * k[p] * k[q];            // g = η + f k k         :291
}

// Stand-in for "a metric function the user wrote" (the reference takes any callable, :302-309): Schwarzschild in
// isotropic coordinates, ds² = −((1−m)/(1+m))² dt² + (1+m)⁴ dx², m = M/(2ρ).  Not of Kerr–Schild form, so it can only go
// through the generic dmetric → christoffel → geodesic chain.  No reference counterpart; the oracle evaluates it for
// metric kind RTGR_USER so that the product's run-time compiled metrics (examples/user_metrics.py) have a checker.
template <class T, class S>
inline S cst(T v) {
    if constexpr (std::is_same<S, T>::value) return v; else return mk<T>(v);
}
template <class T, class S>
inline void schwarzschild_isotropic(const S xx[D], T M, S g[D][D]) {
    S rho = sqrt(pow2(xx[1]) + pow2(xx[2]) + pow2(xx[3]));
    S m = cst<T, S>(T(0.5) * M) / rho;
    S one = cst<T, S>(T(1));
    S lapse = (one - m) / (one + m);
    S psi2 = pow2(one + m);
    for (int p = 0; p < D; p++)
        for (int q = 0; q < D; q++) g[p][q] = cst<T, S>(T(0));
    g[0][0] = cst<T, S>(T(0)) - pow2(lapse);
    g[1][1] = g[2][2] = g[3][3] = pow2(psi2);
}

// A second user-metric stand-in ("zoo"): a smooth, static, non-diagonal perturbation of flat space that goes through EVERY
// elementary function above — the checker of examples/user_metrics.py:HELPER_ZOO.  Selected by
// rtgr_scene.user_metric == RTGR_ORACLE_ZOO (the oracle has no module ids; tests pass the marker in that field).
constexpr uint64_t RTGR_ORACLE_ZOO = 0x200;
template <class T, class S>
inline void helper_zoo(const S xx[D], T M, S g[D][D]) {
    const S &x = xx[1], &y = xx[2], &z = xx[3];
    S rho = sqrt(pow2(x) + pow2(y) + pow2(z));
    S cth = z / rho;
    S th = acos(cth), ph = atan2(y, x), lat = asin(cth);
    S m = cst<T, S>(M);
    for (int p = 0; p < D; p++)
        for (int q = 0; q < D; q++) g[p][q] = cst<T, S>(T(0));
    g[0][0] = cst<T, S>(T(-1)) - (T(0.1) * m) * powr(rho, T(-1.5));
    g[1][1] = cst<T, S>(T(1)) + T(0.05) * abs(lat) + T(0.02) * exp(cst<T, S>(T(0)) - rho);
    g[2][2] = cst<T, S>(T(1)) + T(0.05) * cbrt(cst<T, S>(T(1)) + rho) / rho;
    g[3][3] = cst<T, S>(T(1)) + T(0.03) * atan(rho) * log(cst<T, S>(T(2)) + rho) / rho;
    g[1][2] = g[2][1] = T(0.01) * sin(ph) * cos(th);
    g[0][3] = g[3][0] = T(0.02) * m * cth / rho;
}

// A third stand-in, TIME DEPENDENT: isotropic Schwarzschild whose spatial part expands, g_ij = e^{2Ht} ψ⁴ δ_ij with H = the
// scene's `a` — the checker of examples/user_metrics.py:EXPANDING_ISOTROPIC.  The reference differentiates the metric
// in all four coordinates and evaluates it at the ray's full 4-position (src/RayTraceGR.jl:302-313, :358-363), so a metric
// that depends on t must be traced with ∂_t g ≠ 0 at the ray's own t.  Selected by rtgr_scene.user_metric == RTGR_ORACLE_EXPANDING.
constexpr uint64_t RTGR_ORACLE_EXPANDING = 0x201;
template <class T, class S>
inline void expanding_isotropic(const S xx[D], T M, T H, S g[D][D]) {
    S rho = sqrt(pow2(xx[1]) + pow2(xx[2]) + pow2(xx[3]));
    S m = cst<T, S>(T(0.5) * M) / rho;
    S one = cst<T, S>(T(1));
    S lapse = (one - m) / (one + m);
    S psi2 = pow2(one + m);
    S s = exp(H * xx[0]);
    for (int p = 0; p < D; p++)
        for (int q = 0; q < D; q++) g[p][q] = cst<T, S>(T(0));
    g[0][0] = cst<T, S>(T(0)) - pow2(lapse);
    g[1][1] = g[2][2] = g[3][3] = pow2(s) * pow2(psi2);
}

template <class T, class S>
inline void metric_eval(const rtgr_scene& sc, const S x[D], S g[D][D]) {
    const uint32_t kind = sc.metric & ~RTGR_METRIC_GENERIC;  // the flag picks a product code path, not a metric
    if (kind == RTGR_MINKOWSKI) minkowski<T, S>(x, g);
    else if (kind == RTGR_USER && sc.user_metric == RTGR_ORACLE_ZOO) helper_zoo<T, S>(x, T(sc.M), g);
    else if (kind == RTGR_USER && sc.user_metric == RTGR_ORACLE_EXPANDING) expanding_isotropic<T, S>(x, T(sc.M), T(sc.a), g);
    else if (kind == RTGR_USER) schwarzschild_isotropic<T, S>(x, T(sc.M), g);
    else kerr_schild<T, S>(x, T(sc.M), T(sc.a), (int)kind, g);
}

// dmetric(metric, x): seed 4 duals with unit eps, one metric call, split        src/RayTraceGR.jl:302-313
template <class T>
inline void dmetric(const rtgr_scene& sc, const T x[D], T g[D][D], T dg[D][D][D]) {
    Dual<T> xdx[D];
    for (int a = 0; a < D; a++) {
        xdx[a] = mk<T>(x[a]);
        xdx[a].eps[a] = T(1);
    }
    Dual<T> gdg[D][D];
    metric_eval<T, Dual<T>>(sc, xdx, gdg);
    for (int a = 0; a < D; a++)
        for (int b = 0; b < D; b++) {
            g[a][b] = gdg[a][b].val;
            for (int c = 0; c < D; c++) dg[a][b][c] = gdg[a][b].eps[c];
        }
}

// inv(::SMatrix{4,4}) — StaticArrays 0.12.3 closed-form adjugate / determinant      
// call sites src/RayTraceGR.jl:323, :470
template <class T>
inline void inv4(const T m[D][D], T o[D][D]) {
    T s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1];
    T s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    T s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3];
    T s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    T s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3];
    T s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    T c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    T c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    T c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    T c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    T c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    T c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    T det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    T id = T(1) / det;
    o[0][0] = (m[1][1] * c5 - m[1][2] * c4 + m[1][3] * c3) * id;
    o[0][1] = (-m[0][1] * c5 + m[0][2] * c4 - m[0][3] * c3) * id;
    o[0][2] = (m[3][1] * s5 - m[3][2] * s4 + m[3][3] * s3) * id;
    o[0][3] = (-m[2][1] * s5 + m[2][2] * s4 - m[2][3] * s3) * id;
    o[1][0] = (-m[1][0] * c5 + m[1][2] * c2 - m[1][3] * c1) * id;
    o[1][1] = (m[0][0] * c5 - m[0][2] * c2 + m[0][3] * c1) * id;
    o[1][2] = (-m[3][0] * s5 + m[3][2] * s2 - m[3][3] * s1) * id;
    o[1][3] = (m[2][0] * s5 - m[2][2] * s2 + m[2][3] * s1) * id;
    o[2][0] = (m[1][0] * c4 - m[1][1] * c2 + m[1][3] * c0) * id;
    o[2][1] = (-m[0][0] * c4 + m[0][1] * c2 - m[0][3] * c0) * id;
    o[2][2] = (m[3][0] * s4 - m[3][1] * s2 + m[3][3] * s0) * id;
    o[2][3] = (-m[2][0] * s4 + m[2][1] * s2 - m[2][3] * s0) * id;
    o[3][0] = (-m[1][0] * c3 + m[1][1] * c1 - m[1][2] * c0) * id;
    o[3][1] = (m[0][0] * c3 - m[0][1] * c1 + m[0][2] * c0) * id;
    o[3][2] = (-m[3][0] * s3 + m[3][1] * s1 - m[3][2] * s0) * id;
    o[3][3] = (m[2][0] * s3 - m[2][1] * s1 + m[2][2] * s0) * id;
}

// christoffel(metric, x): all 64 entries, symmetry unused, as written             src/RayTraceGR.jl:321-331
template <class T>
inline void christoffel(const rtgr_scene& sc, const T x[D], T Gam[D][D][D]) {
    T g[D][D], dg[D][D][D], gu[D][D], Gl[D][D][D];
    dmetric<T>(sc, x, g, dg);                                                       // :322
    inv4<T>(g, gu);                                                                 // :323
    for (int a = 0; a < D; a++)
        for (int b = 0; b < D; b++)
            for (int c = 0; c < D; c++) Gl[a][b][c] = (dg[a][b][c] + dg[a][c][b] - dg[b][c][a]) / T(2);  // :324
    for (int a = 0; a < D; a++)
        for (int b = 0; b < D; b++)
            for (int c = 0; c < D; c++)                                              // :326-329
                Gam[a][b][c] = gu[a][0] * Gl[0][b][c] + gu[a][1] * Gl[1][b][c] + gu[a][2] * Gl[2][b][c] +
                               gu[a][3] * Gl[3][b][c];
}

// geodesic(s, metric, λ): (ẋ, u̇) = (u, -Γ^a_xy u^x u^y)                          src/RayTraceGR.jl:358-370
template <class T>
inline void geodesic(const rtgr_scene& sc, const T s[8], T ds[8]) {
    T Gam[D][D][D];
    christoffel<T>(sc, s, Gam);                                                     // :359 (x = s[0..3])
    const T* u = s + D;
    for (int a = 0; a < D; a++) ds[a] = u[a];                                       // :360
    for (int a = 0; a < D; a++) {
        T acc = T(0);
        for (int y = 0; y < D; y++)                                                 // column-major sum  :361-363
            for (int x = 0; x < D; x++) acc += Gam[a][x][y] * u[x] * u[y];
        ds[D + a] = -acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Objects                                                                          src/RayTraceGR.jl:374-441
// ---------------------------------------------------------------------------------------------------------------
template <class T>
inline T sign_of(T v) { return v > T(0) ? T(1) : (v < T(0) ? T(-1) : v); }  // Julia sign(): sign(0)=0, NaN→NaN

template <class T>
inline T distance(const rtgr_object& o, const T pos[D]) {
    switch (o.kind) {
        case RTGR_PLANE:                                                            // pos[1] - pl.time   :399-401
            return pos[0] - T(o.p[0]);
        case RTGR_SPHERE: {                                                         // :415-419
            T R = T(o.p[8]);
            T acc = T(0);
            for (int a = 1; a < D; a++) {
                T d = pos[a] - T(o.p[a]);
                acc += d * d;
            }
            return sign_of(R) * (acc - R * R);
        }
        case RTGR_DISK: {  // no reference counterpart; obeys the contract of :377-383
            T h = T(o.p[0]), rin = T(o.p[1]), rout = T(o.p[2]);
            T rc = std::sqrt(pos[1] * pos[1] + pos[2] * pos[2]);
            T d = std::fabs(pos[3]) - h;
            d = std::fmax(d, rin - rc);
            d = std::fmax(d, rc - rout);
            return d;
        }
        case RTGR_USER_OBJECT:   // new subtypes of the reference's open `abstract type Object{T}` (:374-389): the oracle's twins of
            switch (o.type) {    // examples/user_objects.py SHAPES — same formulas, written against the contract of :377-383
                case 0: {        // torus around the z axis: centre p[0..2], major radius p[3], minor radius p[4]
                    T X = pos[1] - T(o.p[0]), Y = pos[2] - T(o.p[1]), Z = pos[3] - T(o.p[2]);
                    T w = std::sqrt(X * X + Y * Y) - T(o.p[3]);
                    return w * w + Z * Z - T(o.p[4]) * T(o.p[4]);
                }
                case 1: {        // ellipsoid: centre p[0..2], semi-axes p[3..5]
                    T X = (pos[1] - T(o.p[0])) / T(o.p[3]), Y = (pos[2] - T(o.p[1])) / T(o.p[4]), Z = (pos[3] - T(o.p[2])) / T(o.p[5]);
                    return X * X + Y * Y + Z * Z - T(1);
                }
            }
            break;
    }
    return std::numeric_limits<T>::infinity();
}

template <class T>
inline T julia_mod1(T x) {  // mod(x, 1) for floats: x - floor(x), result in [0,1)
    T r = x - std::floor(x);
    if (r >= T(1)) r = T(0);
    return r;
}

template <class T>
inline void objcolor(const rtgr_object& o, const T pos[D], T col[3]) {
    const T pi = T(3.14159265358979323846264338327950288L);
    switch (o.kind) {
        case RTGR_PLANE:                                                            // (0, 1/2, 0)        :402-404
            col[0] = T(0); col[1] = T(1) / T(2); col[2] = T(0);
            return;
        case RTGR_SPHERE: {                                                         // :420-428
            T x = pos[1] - T(o.p[1]), y = pos[2] - T(o.p[2]), z = pos[3] - T(o.p[3]);
            T r = std::sqrt(x * x + y * y + z * z);
            T th = std::acos(z / r);
            T ph = std::atan2(y, x);
            col[0] = julia_mod1(T(12) * th / pi);
            col[1] = julia_mod1(T(12) * ph / pi);
            col[2] = T(1);
            return;
        }
        case RTGR_DISK: {  // checkerboard in (radius, azimuth); no reference counterpart
            T rc = std::sqrt(pos[1] * pos[1] + pos[2] * pos[2]);
            T ph = std::atan2(pos[2], pos[1]);
            col[0] = T(1);
            col[1] = julia_mod1(rc);
            col[2] = julia_mod1(T(12) * ph / pi);
            return;
        }
        case RTGR_USER_OBJECT:
            switch (o.type) {
                case 0: {   // torus: toroidal and poloidal angle
                    T X = pos[1] - T(o.p[0]), Y = pos[2] - T(o.p[1]), Z = pos[3] - T(o.p[2]);
                    T w = std::sqrt(X * X + Y * Y) - T(o.p[3]);
                    col[0] = julia_mod1(T(6) * std::atan2(Y, X) / pi);
                    col[1] = julia_mod1(T(6) * std::atan2(Z, w) / pi);
                    col[2] = T(1) / T(2);
                    return;
                }
                case 1: {   // ellipsoid: the sphere's rule (:420-428) in the scaled coordinates
                    T X = (pos[1] - T(o.p[0])) / T(o.p[3]), Y = (pos[2] - T(o.p[1])) / T(o.p[4]), Z = (pos[3] - T(o.p[2])) / T(o.p[5]);
                    T r = std::sqrt(X * X + Y * Y + Z * Z);
                    col[0] = julia_mod1(T(12) * std::acos(Z / r) / pi);
                    col[1] = T(1) / T(2);
                    col[2] = julia_mod1(T(12) * std::atan2(Y, X) / pi);
                    return;
                }
            }
            break;
    }
    col[0] = col[1] = col[2] = T(0);
}

// objs[o] of `objs::Vector{Object{T}}` (any length, :433-441, :483): the scene's inline slots, or the caller array rtgr_scene.objects
inline const rtgr_object& object_of(const rtgr_scene& sc, uint32_t o) { return sc.objects ? sc.objects[o] : sc.obj[o]; }

// min_distance(objs, s)                                                            src/RayTraceGR.jl:433-441
template <class T>
inline T min_distance(const rtgr_scene& sc, const T x[D]) {
    T dmin = std::numeric_limits<T>::infinity();
    for (uint32_t o = 0; o < sc.nobj; o++) {
        T d = distance<T>(object_of(sc, o), x);
        dmin = (d < dmin || std::isnan(d)) ? d : dmin;  // Julia min() propagates NaN
    }
    return dmin;
}

// ---------------------------------------------------------------------------------------------------------------
// make_canvas                                                                       src/RayTraceGR.jl:457-478
// ---------------------------------------------------------------------------------------------------------------
template <class T>
inline void make_pixel(const rtgr_scene& sc, const rtgr_camera& cam, uint64_t ni, uint64_t nj, uint64_t i0,
                       uint64_t j0, T s[8]) {
    T dx = (T(i0 + 1) - T(1) / T(2)) / T(ni) - T(1) / T(2);                         // :465 (i is 1-based)
    T dy = (T(j0 + 1) - T(1) / T(2)) / T(nj) - T(1) / T(2);                         // :466
    T x[D], n[D], g[D][D], gu[D][D];
    for (int a = 0; a < D; a++) {
        x[a] = T(cam.pos[a]) + dx * T(cam.widthx[a]) + dy * T(cam.widthy[a]);       // :467
        n[a] = T(cam.normal[a]) + dx * T(cam.widthx[a]) + dy * T(cam.widthy[a]);    // :468
    }
    metric_eval<T, T>(sc, x, g);                                                    // :469
    inv4<T>(g, gu);                                                                 // :470
    T t[D];
    for (int a = 0; a < D; a++) t[a] = gu[a][0];                                    // gu * e_t            :471
    T t2 = T(0), n2 = T(0);
    for (int a = 0; a < D; a++)
        for (int b = 0; b < D; b++) {
            t2 += t[a] * g[a][b] * t[b];                                            // :472
            n2 += n[a] * g[a][b] * n[b];                                            // :473
        }
    T st = std::sqrt(-t2), sn = std::sqrt(n2), s2 = std::sqrt(T(2));
    for (int a = 0; a < D; a++) {
        s[a] = x[a];
        s[D + a] = (t[a] / st + n[a] / sn) / s2;                                    // :474
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tsit5 (OrdinaryDiffEq 5.38.3) — constants from SURVEY App. A (verified there against the order conditions)
// ---------------------------------------------------------------------------------------------------------------
template <class T>
struct Tsit5 {
    static constexpr T c1 = T(0.161L), c2 = T(0.327L), c3 = T(0.9L), c4 = T(0.9800255409045097L);
    static constexpr T a21 = T(0.161L);
    static constexpr T a31 = T(-0.008480655492356989L), a32 = T(0.335480655492357L);
    static constexpr T a41 = T(2.8971530571054935L), a42 = T(-6.359448489975075L), a43 = T(4.3622954328695815L);
    static constexpr T a51 = T(5.325864828439257L), a52 = T(-11.748883564062828L), a53 = T(7.4955393428898365L),
                       a54 = T(-0.09249506636175525L);
    static constexpr T a61 = T(5.86145544294642L), a62 = T(-12.92096931784711L), a63 = T(8.159367898576159L),
                       a64 = T(-0.071584973281401L), a65 = T(-0.028269050394068383L);
    static constexpr T a71 = T(0.09646076681806523L), a72 = T(0.01L), a73 = T(0.4798896504144996L),
                       a74 = T(1.379008574103742L), a75 = T(-3.290069515436081L), a76 = T(2.324710524099774L);
    static constexpr T bt1 = T(-0.00178001105222577714L), bt2 = T(-0.0008164344596567469L),
                       bt3 = T(0.007880878010261995L), bt4 = T(-0.1447110071732629L), bt5 = T(0.5823571654525552L),
                       bt6 = T(-0.45808210592918697L), bt7 = T(0.015151515151515152L);
    // dense output b_i(θ) = θ r_i1 + θ² r_i2 + θ³ r_i3 + θ⁴ r_i4
    static constexpr T r11 = T(1.0L), r12 = T(-2.763706197274826L), r13 = T(2.9132554618219126L),
                       r14 = T(-1.0530884977290216L);
    static constexpr T r22 = T(0.13169999999999998L), r23 = T(-0.2234L), r24 = T(0.1017L);
    static constexpr T r32 = T(3.9302962368947516L), r33 = T(-5.941033872131505L), r34 = T(2.490627285651252793L);
    static constexpr T r42 = T(-12.411077166933676L), r43 = T(30.33818863028232L), r44 = T(-16.548102889244902L);
    static constexpr T r52 = T(37.50931341651104L), r53 = T(-88.1789048947664L), r54 = T(47.37952196281928L);
    static constexpr T r62 = T(-27.896526289197286L), r63 = T(65.09189467479366L), r64 = T(-34.87065786149661L);
    static constexpr T r72 = T(1.5L), r73 = T(-4.0L), r74 = T(2.5L);
};

// ODE_DEFAULT_NORM for an 8-vector: sqrt(sum(abs2,u)/length(u))                       
template <class T>
inline T rms8(const T v[8]) {
    T acc = T(0);
    for (int i = 0; i < 8; i++) acc += v[i] * v[i];
    return std::sqrt(acc / T(8));
}

// dense output of one step: u(t + θ h) = y0 + h Σ b_i(θ) k_i      (Tsit5 ode_interpolant)
template <class T>
inline void dense8(const T y0[8], T h, const T k[7][8], T th, T out[8]) {
    using C = Tsit5<T>;
    T th2 = th * th;
    T b1 = th * (C::r11 + th * (C::r12 + th * (C::r13 + th * C::r14)));
    T b2 = th2 * (C::r22 + th * (C::r23 + th * C::r24));
    T b3 = th2 * (C::r32 + th * (C::r33 + th * C::r34));
    T b4 = th2 * (C::r42 + th * (C::r43 + th * C::r44));
    T b5 = th2 * (C::r52 + th * (C::r53 + th * C::r54));
    T b6 = th2 * (C::r62 + th * (C::r63 + th * C::r64));
    T b7 = th2 * (C::r72 + th * (C::r73 + th * C::r74));
    for (int i = 0; i < 8; i++)
        out[i] = y0[i] + h * (k[0][i] * b1 + k[1][i] * b2 + k[2][i] * b3 + k[3][i] * b4 + k[4][i] * b5 +
                              k[5][i] * b6 + k[6][i] * b7);
}

struct RayResult {
    uint8_t status;
    uint8_t interior;  // event found by an interior sample point
    uint32_t nacc, nrej, nrhs;
};

// One trajectory: solve(ODEProblem(geodesic, s0, (λ0, λ1), metric), Tsit5(), callback=cb, reltol, abstol)
// call site src/RayTraceGR.jl:497-511; semantics SURVEY App. B.1-B.4
template <class T>
RayResult solve_ray(const rtgr_scene& sc, const rtgr_solver& opt, const T s0[8], T s_end[8], T* lam_end) {
    using C = Tsit5<T>;
    const T reltol = T(opt.reltol), abstol = T(opt.abstol);
    const T t0 = T(opt.lambda0), t1 = T(opt.lambda1);
    const T dtmax = t1 - t0;
    const T beta1 = T(7) / T(50), beta2 = T(2) / T(25), gamma = T(9) / T(10);
    const T qmin = T(1) / T(5), qmax = T(10), qoldinit = T(1) / T(10000);
    const T eps = std::numeric_limits<T>::epsilon();
    RayResult res{RTGR_RAY_LAMBDA1, 0, 0, 0, 0};

    T y[8];
    for (int i = 0; i < 8; i++) y[i] = s0[i];
    T t = t0;
    T k[7][8];

    // ---- initial dt (Hairer), SURVEY App. B.3 -----------------------------------------------------------
    T f0[8];
    geodesic<T>(sc, y, f0);
    res.nrhs++;
    T dt;
    {
        T sk[8], tmp[8];
        for (int i = 0; i < 8; i++) sk[i] = abstol + std::fabs(y[i]) * reltol;
        for (int i = 0; i < 8; i++) tmp[i] = y[i] / sk[i];
        T d0 = rms8(tmp);
        for (int i = 0; i < 8; i++) tmp[i] = f0[i] / sk[i];
        T d1 = rms8(tmp);
        T dt0 = (d0 < T(1e-5) || d1 < T(1e-5)) ? T(1e-6) : (d0 / d1) / T(100);
        dt0 = std::fmin(dt0, dtmax);
        T u1[8], f1[8];
        for (int i = 0; i < 8; i++) u1[i] = y[i] + dt0 * f0[i];
        geodesic<T>(sc, u1, f1);
        res.nrhs++;
        for (int i = 0; i < 8; i++) tmp[i] = (f1[i] - f0[i]) / sk[i];
        T d2 = rms8(tmp) / dt0;
        T md = std::fmax(d1, d2);
        T dt1;
        if (md <= T(1e-15)) dt1 = std::fmax(T(1e-6), dt0 * T(1e-3));
        else dt1 = std::pow(T(10), -(T(2) + std::log10(md)) / T(5));
        dt = std::fmin(std::fmin(T(100) * dt0, dt1), dtmax);
    }
    for (int i = 0; i < 8; i++) k[0][i] = f0[i];  // fsalfirst
    T qold = qoldinit;
    const uint32_t maxsteps = opt.max_steps;
    const int npts = (int)opt.interp_points;

    while (true) {
        if (!(t < t1)) { res.status = RTGR_RAY_LAMBDA1; break; }
        if (res.nacc + res.nrej >= maxsteps) { res.status = RTGR_RAY_MAXSTEPS; break; }
        // tstop clipping (modify_dt_for_tstops!), SURVEY App. B.2
        if (dt > t1 - t) dt = t1 - t;
        // ---- one Tsit5 attempt, SURVEY App. B.1 ---------------------------------------------------------
        T Y[8], ynew[8];
        for (int i = 0; i < 8; i++) Y[i] = y[i] + (dt * C::a21) * k[0][i];
        geodesic<T>(sc, Y, k[1]);
        for (int i = 0; i < 8; i++) Y[i] = y[i] + dt * (C::a31 * k[0][i] + C::a32 * k[1][i]);
        geodesic<T>(sc, Y, k[2]);
        for (int i = 0; i < 8; i++) Y[i] = y[i] + dt * (C::a41 * k[0][i] + C::a42 * k[1][i] + C::a43 * k[2][i]);
        geodesic<T>(sc, Y, k[3]);
        for (int i = 0; i < 8; i++)
            Y[i] = y[i] + dt * (C::a51 * k[0][i] + C::a52 * k[1][i] + C::a53 * k[2][i] + C::a54 * k[3][i]);
        geodesic<T>(sc, Y, k[4]);
        for (int i = 0; i < 8; i++)
            Y[i] = y[i] + dt * (C::a61 * k[0][i] + C::a62 * k[1][i] + C::a63 * k[2][i] + C::a64 * k[3][i] +
                                C::a65 * k[4][i]);
        geodesic<T>(sc, Y, k[5]);
        for (int i = 0; i < 8; i++)
            ynew[i] = y[i] + dt * (C::a71 * k[0][i] + C::a72 * k[1][i] + C::a73 * k[2][i] + C::a74 * k[3][i] +
                                   C::a75 * k[4][i] + C::a76 * k[5][i]);
        geodesic<T>(sc, ynew, k[6]);
        res.nrhs += 6;
        T resid[8];
        for (int i = 0; i < 8; i++) {
            T ut = dt * (C::bt1 * k[0][i] + C::bt2 * k[1][i] + C::bt3 * k[2][i] + C::bt4 * k[3][i] +
                         C::bt5 * k[4][i] + C::bt6 * k[5][i] + C::bt7 * k[6][i]);
            resid[i] = ut / (abstol + std::fmax(std::fabs(y[i]), std::fabs(ynew[i])) * reltol);
        }
        T EEst = rms8(resid);
        if (std::isnan(EEst)) { res.status = RTGR_RAY_NAN; break; }
        // ---- PI controller, SURVEY App. B.2 -------------------------------------------------------------
        T q, q11 = T(0);
        if (EEst == T(0)) q = T(1) / qmax;
        else {
            q11 = std::pow(EEst, beta1);
            q = q11 / std::pow(qold, beta2);
            q = std::fmax(T(1) / qmax, std::fmin(T(1) / qmin, q / gamma));
        }
        if (EEst <= T(1)) {
            // accept
            res.nacc++;
            qold = std::fmax(EEst, qoldinit);
            T dtnew = dt / q;
            T tprev = t;
            T tnew = t + dt;
            if (std::fabs(tnew - t1) < T(10) * eps * std::fmax(std::fabs(tnew), std::fabs(t1))) tnew = t1;
            // ---- ContinuousCallback, SURVEY App. B.4 ----------------------------------------------------
            T prev_sign = sign_of(min_distance<T>(sc, y));
            T next_sign = sign_of(min_distance<T>(sc, ynew));
            bool event = false;
            T top = T(1);
            if (prev_sign != T(0) && prev_sign * next_sign <= T(0)) {
                event = true;
            } else if (prev_sign != T(0) && npts > 1) {
                for (int i = 2; i <= npts; i++) {
                    T th = T(i - 1) / T(npts - 1);
                    T yi[8];
                    dense8<T>(y, dt, k, th, yi);
                    T sg = sign_of(min_distance<T>(sc, yi));
                    if (prev_sign * sg < T(0)) {
                        event = true;
                        top = th;
                        if (i != npts) res.interior = 1;
                        break;
                    }
                }
            }
            if (event) {
                // bracketed root of cond(dense(θ)) on [0, top]; Θ stays on the pre-crossing side
                // (Roots' Alefeld–Potra–Shi + prevfloat in the reference; bisection to the last ulp here)
                T lo = T(0), hi = top;
                T yi[8];
                dense8<T>(y, dt, k, hi, yi);
                T Theta;
                if (min_distance<T>(sc, yi) == T(0)) {
                    Theta = hi;
                } else {
                    for (int it = 0; it < 200; it++) {
                        T mid = lo + (hi - lo) / T(2);
                        if (!(mid > lo && mid < hi)) break;
                        dense8<T>(y, dt, k, mid, yi);
                        T sg = sign_of(min_distance<T>(sc, yi));
                        if (sg == T(0)) { hi = mid; continue; }
                        if (sg * prev_sign > T(0)) lo = mid; else hi = mid;
                    }
                    Theta = lo;
                }
                dense8<T>(y, dt, k, Theta, s_end);
                *lam_end = tprev + dt * Theta;
                res.status = RTGR_RAY_EVENT;
                return res;
            }
            for (int i = 0; i < 8; i++) { y[i] = ynew[i]; k[0][i] = k[6][i]; }  // FSAL
            t = tnew;
            dt = std::fmin(dtmax, dtnew);
            if (!(dt > T(0)) || t + dt == t) {
                if (t < t1) { res.status = RTGR_RAY_DTMIN; break; }
            }
        } else {
            // reject
            res.nrej++;
            dt = dt / std::fmin(T(1) / qmin, q11 / gamma);
            if (t + dt == t) { res.status = RTGR_RAY_DTMIN; break; }
        }
    }
    for (int i = 0; i < 8; i++) s_end[i] = y[i];
    *lam_end = t;
    return res;
}

// colouring loop of trace_rays                                                     src/RayTraceGR.jl:513-533
template <class T>
inline uint32_t colour(const rtgr_scene& sc, const rtgr_solver& opt, const T x[D], T col[3]) {
    uint32_t omin = 0;
    T dmin = T(opt.hit_threshold);                                                  // :519
    for (uint32_t o = 0; o < sc.nobj; o++) {                                        // :520-526
        T d = distance<T>(object_of(sc, o), x);
        if (d < dmin) { omin = o + 1; dmin = d; }
    }
    if (omin == 0) {                                                                // :527-528
        for (int c = 0; c < 3; c++) col[c] = T(opt.miss_rgb[c]);
    } else {                                                                        // :530
        objcolor<T>(object_of(sc, omin - 1), x, col);
        T scale = T(omin) / T(sc.nobj);
        for (int c = 0; c < 3; c++) col[c] *= scale;
    }
    return omin;
}

template <class T>
int trace_impl(const rtgr_scene* sc, const rtgr_solver* opt, const T* state0, const rtgr_camera* cam, uint64_t ni,
               uint64_t nj, uint64_t j0, uint64_t j1, T* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr,
               int nthreads) {
    if (!sc || !opt || !rgb || j1 <= j0 || j1 > nj || ni == 0 || (sc->nobj > RTGR_MAX_OBJECTS && !sc->objects)) return RTGR_ERR_BAD_ARG;
    if (!state0 && !cam) return RTGR_ERR_BAD_ARG;
    const uint64_t n = ni * (j1 - j0);
    uint64_t acc = 0, rej = 0, rhs = 0, ev = 0, evi = 0, nf = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) reduction(+ : acc, rej, rhs, ev, evi, nf)
#endif
    for (int64_t idx = 0; idx < (int64_t)n; idx++) {
        T s0[8], se[8], lam, col[3];
        if (state0) for (int a = 0; a < 8; a++) s0[a] = state0[idx * 8 + a];
        else make_pixel<T>(*sc, *cam, ni, nj, (uint64_t)idx % ni, j0 + (uint64_t)idx / ni, s0);
        RayResult r = solve_ray<T>(*sc, *opt, s0, se, &lam);
        const uint32_t hit = colour<T>(*sc, *opt, se, col);
        for (int c = 0; c < 3; c++) rgb[(uint64_t)c * n + idx] = col[c];
        if (out) {
            if (out->state_end) for (int a = 0; a < 8; a++) ((T*)out->state_end)[idx * 8 + a] = se[a];
            if (out->lambda_end) ((T*)out->lambda_end)[idx] = lam;
            if (out->status) out->status[idx] = r.status;
            if (out->hit) out->hit[idx] = (uint8_t)hit;
            if (out->hit32) out->hit32[idx] = hit;
            if (out->n_accept) out->n_accept[idx] = r.nacc;
            if (out->n_reject) out->n_reject[idx] = r.nrej;
        }
        acc += r.nacc; rej += r.nrej; rhs += r.nrhs;
        ev += (r.status == RTGR_RAY_EVENT); evi += r.interior; nf += (r.status >= RTGR_RAY_MAXSTEPS);
    }
    if (ctr) {
        ctr->rays = n; ctr->accepted = acc; ctr->rejected = rej; ctr->rhs_evals = rhs;
        ctr->events = ev; ctr->events_interior = evi; ctr->not_finished = nf; ctr->reserved = 0;
    }
    return RTGR_OK;
}

// distance(obj, x) of every object, min_distance and the colour rule at n points: the checker of rtgr_eval_objects_f64 / _f32
template <class T>
int eval_objects_impl(const rtgr_scene* sc, const rtgr_solver* opt, const T* x, uint64_t n, T* d, T* dmin, uint8_t* hit, T* rgb) {
    for (uint64_t p = 0; p < n; p++) {
        const T* xp = x + 4 * p;
        if (d) for (uint32_t o = 0; o < sc->nobj; o++) d[p * sc->nobj + o] = distance<T>(object_of(*sc, o), xp);   // :377-419
        if (dmin) dmin[p] = min_distance<T>(*sc, xp);                                                        // :433-441
        T col[3];
        const uint32_t h = colour<T>(*sc, *opt, xp, col);                                                    // :513-533
        if (hit) hit[p] = (uint8_t)h;
        if (rgb) for (int c = 0; c < 3; c++) rgb[3 * p + c] = col[c];
    }
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C entry points (ctypes from tests/ and bench.py's cpu_baseline only)
// ---------------------------------------------------------------------------------------------------------------
extern "C" {

int rtgr_oracle_trace_f64(const rtgr_scene* sc, const rtgr_solver* opt, const double* state0, const rtgr_camera* cam,
                          uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* rgb,
                          const rtgr_ray_outputs* out, rtgr_counters* ctr, int nthreads) {
    return trace_impl<double>(sc, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr, nthreads);
}
int rtgr_oracle_trace_f32(const rtgr_scene* sc, const rtgr_solver* opt, const float* state0, const rtgr_camera* cam,
                          uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* rgb,
                          const rtgr_ray_outputs* out, rtgr_counters* ctr, int nthreads) {
    return trace_impl<float>(sc, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr, nthreads);
}
int rtgr_oracle_make_canvas_f64(const rtgr_scene* sc, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                                uint64_t j1, double* state0) {
    if (!sc || !cam || !state0 || j1 <= j0 || j1 > nj) return RTGR_ERR_BAD_ARG;
    for (uint64_t j = j0; j < j1; j++)
        for (uint64_t i = 0; i < ni; i++) make_pixel<double>(*sc, *cam, ni, nj, i, j, state0 + (i + (j - j0) * ni) * 8);
    return RTGR_OK;
}
int rtgr_oracle_make_canvas_f32(const rtgr_scene* sc, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                                uint64_t j1, float* state0) {
    if (!sc || !cam || !state0 || j1 <= j0 || j1 > nj) return RTGR_ERR_BAD_ARG;
    for (uint64_t j = j0; j < j1; j++)
        for (uint64_t i = 0; i < ni; i++) make_pixel<float>(*sc, *cam, ni, nj, i, j, state0 + (i + (j - j0) * ni) * 8);
    return RTGR_OK;
}
// g (n x 16), dg (n x 64, [a][b][c]), Gamma (n x 64); any output may be NULL
int rtgr_oracle_eval_metric_f64(const rtgr_scene* sc, const double* x, uint64_t n, double* g, double* dg,
                                double* Gam) {
    if (!sc || !x) return RTGR_ERR_BAD_ARG;
    for (uint64_t p = 0; p < n; p++) {
        double gg[D][D], dd[D][D][D], GG[D][D][D];
        dmetric<double>(*sc, x + 4 * p, gg, dd);
        if (g) std::memcpy(g + 16 * p, gg, sizeof gg);
        if (dg) std::memcpy(dg + 64 * p, dd, sizeof dd);
        if (Gam) {
            christoffel<double>(*sc, x + 4 * p, GG);
            std::memcpy(Gam + 64 * p, GG, sizeof GG);
        }
    }
    return RTGR_OK;
}
int rtgr_oracle_eval_metric_f32(const rtgr_scene* sc, const float* x, uint64_t n, float* g, float* dg, float* Gam) {
    if (!sc || !x) return RTGR_ERR_BAD_ARG;
    for (uint64_t p = 0; p < n; p++) {
        float gg[D][D], dd[D][D][D], GG[D][D][D];
        dmetric<float>(*sc, x + 4 * p, gg, dd);
        if (g) std::memcpy(g + 16 * p, gg, sizeof gg);
        if (dg) std::memcpy(dg + 64 * p, dd, sizeof dd);
        if (Gam) {
            christoffel<float>(*sc, x + 4 * p, GG);
            std::memcpy(Gam + 64 * p, GG, sizeof GG);
        }
    }
    return RTGR_OK;
}
// plain metric call metric(x) (no duals), for the reference's `g - metric(x)` check (test/runtests.jl:57)
int rtgr_oracle_metric_plain_f64(const rtgr_scene* sc, const double* x, uint64_t n, double* g) {
    if (!sc || !x || !g) return RTGR_ERR_BAD_ARG;
    for (uint64_t p = 0; p < n; p++) {
        double gg[D][D];
        metric_eval<double, double>(*sc, x + 4 * p, gg);
        std::memcpy(g + 16 * p, gg, sizeof gg);
    }
    return RTGR_OK;
}
int rtgr_oracle_inv4_f64(const double* m, double* o) {
    double a[D][D], b[D][D];
    std::memcpy(a, m, sizeof a);
    inv4<double>(a, b);
    std::memcpy(o, b, sizeof b);
    return RTGR_OK;
}
int rtgr_oracle_eval_geodesic_f64(const rtgr_scene* sc, const double* s, uint64_t n, double* ds) {
    if (!sc || !s || !ds) return RTGR_ERR_BAD_ARG;
    for (uint64_t p = 0; p < n; p++) geodesic<double>(*sc, s + 8 * p, ds + 8 * p);
    return RTGR_OK;
}
// high-precision cross-check of the RHS: the same as-written chain in long double
int rtgr_oracle_eval_geodesic_ld(const rtgr_scene* sc, const double* s, uint64_t n, double* ds) {
    if (!sc || !s || !ds) return RTGR_ERR_BAD_ARG;
    for (uint64_t p = 0; p < n; p++) {
        long double si[8], so[8];
        for (int i = 0; i < 8; i++) si[i] = s[8 * p + i];
        geodesic<long double>(*sc, si, so);
        for (int i = 0; i < 8; i++) ds[8 * p + i] = (double)so[i];
    }
    return RTGR_OK;
}
// redshift g = (k.u_obs)/(k.u_emit) as include/rtgr.h (rtgr_ray_outputs.redshift) defines it — NO reference counterpart
// (Sphere.vel is stored and never used, src/RayTraceGR.jl:411, :416): the oracle restates the DEFINITION independently,
// with its own metric and its own 4x4 inverse.
int rtgr_oracle_redshift_f64(const rtgr_scene* sc, const double* state0, const double* state_end, const uint8_t* hit,
                             uint64_t n, double* out) {
    if (!sc || !state0 || !state_end || !hit || !out) return RTGR_ERR_BAD_ARG;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    auto observer = [](const double g[D][D], double t[D]) {
        double gu[D][D];
        inv4<double>(g, gu);
        double t2 = 0;
        for (int p = 0; p < D; p++) t[p] = gu[p][0];
        for (int p = 0; p < D; p++) for (int q = 0; q < D; q++) t2 += t[p] * g[p][q] * t[q];
        if (!(t2 < 0)) return false;
        for (int p = 0; p < D; p++) t[p] /= -std::sqrt(-t2);   // future-directed (g^{-1} e_t itself points to the past)
        return true;
    };
    auto dot = [](const double g[D][D], const double* a, const double* b) {
        double s = 0;
        for (int p = 0; p < D; p++) for (int q = 0; q < D; q++) s += a[p] * g[p][q] * b[q];
        return s;
    };
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t h = hit[i];
        if (h == 0 || h > sc->nobj) { out[i] = nan; continue; }
        double g0[D][D], ge[D][D], uo[D], ue[D];
        metric_eval<double, double>(*sc, state0 + 8 * i, g0);
        metric_eval<double, double>(*sc, state_end + 8 * i, ge);
        bool ok = observer(g0, uo);
        const rtgr_object& ob = object_of(*sc, h - 1);
        if (ob.kind == RTGR_SPHERE) {
            const double* v = ob.p + 4;
            const double v2 = dot(ge, v, v);
            ok = ok && v2 < 0;
            for (int p = 0; p < D; p++) ue[p] = v[p] / std::sqrt(-v2);
        } else {
            ok = observer(ge, ue) && ok;
        }
        out[i] = ok ? dot(g0, state0 + 8 * i + 4, uo) / dot(ge, state_end + 8 * i + 4, ue) : nan;
    }
    return RTGR_OK;
}
int rtgr_oracle_eval_objects_f64(const rtgr_scene* sc, const rtgr_solver* opt, const double* x, uint64_t n, double* d, double* dmin, uint8_t* hit, double* rgb) {
    return eval_objects_impl<double>(sc, opt, x, n, d, dmin, hit, rgb);
}
int rtgr_oracle_eval_objects_f32(const rtgr_scene* sc, const rtgr_solver* opt, const float* x, uint64_t n, float* d, float* dmin, uint8_t* hit, float* rgb) {
    return eval_objects_impl<float>(sc, opt, x, n, d, dmin, hit, rgb);
}

int rtgr_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
}
