// rtgr_physics.hpp — device-side physics of the geodesic hot path (gfx950 / CDNA4, wave64).
//
// What the reference computes per RHS evaluation (src/RayTraceGR.jl:358-370 -> :321-331 -> :302-313 -> :274-294):
// forward-mode duals through the metric, a 4x4 inverse, 64 Christoffel symbols, and the contraction
// u̇^a = -Γ^a_bc u^b u^c.  Two device formulations live here:
//
//  * ksform_rhs  (production): both built-in metrics are of Kerr–Schild form g = η + f k⊗k with k_t = 1 and no
//    t-dependence.  The Jacobian of (f, k_i) is propagated in registers by the chain rule through the two
//    intermediates (r, z) — the same first-order information the reference's Dual{T,SVector{4}} carries — and the
//    Christoffel contraction is done BEFORE raising, using the Sherman–Morrison inverse
//    g^{-1} = η^{-1} - f k♯k♯ / (1 + f k·η^{-1}·k).  No 4x4x4 array is ever formed.
//  * generic_rhs (validation / API parity): 4-wide forward duals through an arbitrary metric functor, symmetric
//    g (10) and dg (40), cofactor inverse, contract-then-raise.  This is the "lean generic-metric" form whose flop
//    count (814) defines the algorithmic work in SURVEY §8(d).  It also feeds rtgr_eval_metric_f64 (g, dg, Γ).
//
// Everything is templated on the scalar R (double for the reference's Float64 path, float for config C4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtgr.h"
#include "rtgr_args.hpp"

namespace rtgr {

#define RTGR_DEV __device__ __forceinline__

template <class T> struct IsF64 { static constexpr bool v = false; };   // (no <type_traits>: the user units are also built by hiprtc)
template <> struct IsF64<double> { static constexpr bool v = true; };

// ---- small math helpers -------------------------------------------------------------------------------------------
template <class R> RTGR_DEV R rfma(R a, R b, R c);
template <> RTGR_DEV double rfma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> RTGR_DEV float rfma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <class R> RTGR_DEV R rsqrt_(R x);
template <> RTGR_DEV double rsqrt_<double>(double x) { return __builtin_sqrt(x); }
template <> RTGR_DEV float rsqrt_<float>(float x) { return __builtin_sqrtf(x); }
template <class R> RTGR_DEV R rabs(R x) { return x < R(0) ? -x : x; }
template <> RTGR_DEV double rabs<double>(double x) { return __builtin_fabs(x); }
template <> RTGR_DEV float rabs<float>(float x) { return __builtin_fabsf(x); }
// v_max_f64 / v_min_f64, ONE instruction each (IEEE maxNum/minNum; operands are never NaN where used).  As inline asm:
// the builtins add a canonicalising v_max x, x per computed operand (28 of the 95 min/max in the NEAR kernel).
template <class R> RTGR_DEV R rmax(R a, R b);
template <> RTGR_DEV double rmax<double>(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <> RTGR_DEV float rmax<float>(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max(|a|, |b|) in ONE instruction (source modifiers).  __builtin_fmax(__builtin_fabs(a), __builtin_fabs(b)) costs three:
// LLVM canonicalises each operand of an IEEE maxnum with a v_max x, x of its own — 16 wasted issue slots per step in
// the error norm alone.
template <class R> RTGR_DEV R rmaxabs(R a, R b);
template <> RTGR_DEV double rmaxabs<double>(double a, double b) {
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <> RTGR_DEV float rmaxabs<float>(float a, float b) {
    float r;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <class R> RTGR_DEV R rmin(R a, R b);
template <> RTGR_DEV double rmin<double>(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <> RTGR_DEV float rmin<float>(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// ---- fast reciprocal / reciprocal square root: hardware seed + ONE third-order correction on the FMA pipe.
// Measured on gfx950 (rtgr_eval_fastmath_f64 and its GPU test, test_fast_reciprocal_and_rsqrt_accuracy; the seeds
// themselves in round 1 with a stand-alone microbenchmark): v_rcp_f64 / v_rsq_f64 seeds are good to 2^-24.4 / 2^-24.2, so a
// cubically convergent step (error e³ ≈ 2^-73) lands on full double precision: max relative error 1.1e-16 / 1.4e-16
// over 2^20 operands in [2^-10, 2^10].  No denormal / inf fix-up (operands here are O(1e-3 … 1e3)).  The IEEE
// expansions hipcc emits for `1.0/x` and `sqrt(x)` cost 11 and ~14 instructions; these cost 4 and 6.
template <class R> RTGR_DEV R frcp(R x);
template <> RTGR_DEV double frcp<double>(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);              // 1 − x r
    return __builtin_fma(r, __builtin_fma(e, e, e), r);      // r (1 + e + e²)
}
template <> RTGR_DEV float frcp<float>(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
template <class R> RTGR_DEV R frsq(R x);  // 1/sqrt(x)
template <> RTGR_DEV double frsq<double>(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * r, r, 1.0);          // 1 − x r²
    return __builtin_fma(r * e, __builtin_fma(0.375, e, 0.5), r);  // r (1 + e/2 + 3e²/8)
}
template <> RTGR_DEV float frsq<float>(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-x * r, r, 1.0f);
    return __builtin_fmaf(0.5f * r, e, r);
}
template <bool FAST, class R> RTGR_DEV R rcp_(R x) { if constexpr (FAST) return frcp<R>(x); else return R(1) / x; }
template <bool FAST, class R> RTGR_DEV R sqrt_(R x) { if constexpr (FAST) return x * frsq<R>(x); else return rsqrt_(x); }

// s = sqrt(x) and is = 1/sqrt(x) from ONE reciprocal-square-root (FAST) or the IEEE pair
template <bool FAST, class R> RTGR_DEV void sqrt_inv(R x, R& s, R& is) {
    if constexpr (FAST) { is = frsq<R>(x); s = x * is; }
    else { s = rsqrt_(x); is = R(1) / s; }
}

// Julia sign(): ±1, 0 -> 0 (NaN handled by callers)
template <class R> RTGR_DEV R rsign(R v) { return v > R(0) ? R(1) : (v < R(0) ? R(-1) : R(0)); }

RTGR_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- Kerr–Schild-form field: f, ∇f, k_i, ∂_j k_i at a spatial point ----------------------------------------------
template <class R>
struct KSField {
    R f, df[3];      // f, ∂_j f
    R k[3];          // k_x, k_y, k_z   (k_t = 1)
    R dk[3][3];      // dk[i][j] = ∂_j k_i
};

// METRIC: RTGR_KS_REF — kerr_schild as written (src/RayTraceGR.jl:283-289, r of :284) — or RTGR_KS_TRUE.
// SPIN = false is the reference's actual configuration (a = 0 hard-wired, :276) and drops the a-terms statically.
template <class R, int METRIC, bool SPIN, bool FAST = false>
RTGR_DEV void ks_field(R x, R y, R z, R M, R a, KSField<R>& F) {
#pragma clang fp contract(off)   // (part of the camera: see make_pixel, rtgr_integrator.hpp)
    const R a2 = SPIN ? a * a : R(0);
    const R rho2 = rfma(x, x, rfma(y, y, z * z));
    R r, rq2, rz;  // r, 2*∂r/∂q (so that ∇r = rq2*(x,y,z) + rz*ẑ), explicit ∂r/∂z
    R ir;  // 1/r
    if constexpr (METRIC == RTGR_KS_REF) {
        if constexpr (SPIN) {
            // r = sqrt(q)/2 + sqrt(a² z² + (q/2)²),  q = ρ² - a²                                     :284
            const R q = rho2 - a2;
            R s1, is1, s2, is2;
            sqrt_inv<FAST>(q, s1, is1);
            const R hq = R(0.5) * q;
            sqrt_inv<FAST>(rfma(a2 * z, z, hq * hq), s2, is2);
            r = rfma(R(0.5), s1, s2);
            ir = rcp_<FAST>(r);
            rq2 = R(0.5) * rfma(q, is2, is1);  // 2*(1/(4 s1) + q/(4 s2))
            rz = a2 * z * is2;
        } else {
            // a = 0: r = ρ/2 + ρ²/2
            R rho, irho;
            sqrt_inv<FAST>(rho2, rho, irho);
            r = R(0.5) * (rho + rho2);
            ir = R(2) * irho * rcp_<FAST>(R(1) + rho);
            rq2 = rfma(R(0.5), irho, R(1));  // ∇r = (1/(2ρ) + 1)(x,y,z)
            rz = R(0);
        }
    } else {
        if constexpr (SPIN) {
            // textbook: r² = (q + sqrt(q² + 4a²z²))/2
            const R q = rho2 - a2;
            R sq, is;
            sqrt_inv<FAST>(rfma(q, q, R(4) * a2 * z * z), sq, is);
            const R r2 = R(0.5) * (q + sq);
            sqrt_inv<FAST>(r2, r, ir);
            rq2 = R(0.5) * rfma(q, is, R(1)) * ir;  // 2 * (1+q/s)/2 / (2r)
            rz = a2 * z * is * ir;
        } else {
            sqrt_inv<FAST>(rho2, r, ir);  // a = 0: r = ρ
            rq2 = ir;
            rz = R(0);
        }
    }
    const R dr[3] = {rq2 * x, rq2 * y, rfma(rq2, z, rz)};
    const R r2 = r * r;
    if constexpr (SPIN) {
        // f = 2M r³/(r⁴ + a² z²)                                                                   :285
        const R a2z = a2 * z;
        const R den = rfma(r2, r2, a2z * z);
        const R iden = rcp_<FAST>(den);
        const R r3 = r2 * r;
        const R tm = R(2) * M;
        F.f = tm * r3 * iden;
        // ∂f/∂r = 2M r²(3 den − 4 r⁴)/den² ;  ∂f/∂z|_r = −2M r³ 2a²z/den²
        const R fr = tm * r2 * rfma(R(3), den, R(-4) * r2 * r2) * iden * iden;
        const R fz = R(-2) * F.f * a2z * iden;
        F.df[0] = fr * dr[0];
        F.df[1] = fr * dr[1];
        F.df[2] = rfma(fr, dr[2], fz);
        // k = ((r x + a y), (r y − a x))/(r² + a²),  z/r                                           :286-289
        const R w = rcp_<FAST>(r2 + a2);
        F.k[0] = rfma(r, x, a * y) * w;
        F.k[1] = rfma(r, y, -a * x) * w;
        F.k[2] = z * ir;
        const R kxr = w * rfma(R(-2) * r, F.k[0], x);  // ∂k_x/∂r
        const R kyr = w * rfma(R(-2) * r, F.k[1], y);
        const R kzr = -F.k[2] * ir;
        const R rw = r * w, aw = a * w;
        F.dk[0][0] = rfma(kxr, dr[0], rw);  F.dk[0][1] = rfma(kxr, dr[1], aw);  F.dk[0][2] = kxr * dr[2];
        F.dk[1][0] = rfma(kyr, dr[0], -aw); F.dk[1][1] = rfma(kyr, dr[1], rw);  F.dk[1][2] = kyr * dr[2];
        F.dk[2][0] = kzr * dr[0];           F.dk[2][1] = kzr * dr[1];           F.dk[2][2] = rfma(kzr, dr[2], ir);
    } else {
        // a = 0: f = 2M/r, k_i = x_i/r
        F.f = R(2) * M * ir;
        const R fr = -F.f * ir;
        F.df[0] = fr * dr[0]; F.df[1] = fr * dr[1]; F.df[2] = fr * dr[2];
        F.k[0] = x * ir; F.k[1] = y * ir; F.k[2] = z * ir;
        const R c0 = -F.k[0] * ir, c1 = -F.k[1] * ir, c2 = -F.k[2] * ir;  // ∂k_i/∂r = −x_i/r²
        F.dk[0][0] = rfma(c0, dr[0], ir); F.dk[0][1] = c0 * dr[1];          F.dk[0][2] = c0 * dr[2];
        F.dk[1][0] = c1 * dr[0];          F.dk[1][1] = rfma(c1, dr[1], ir); F.dk[1][2] = c1 * dr[2];
        F.dk[2][0] = c2 * dr[0];          F.dk[2][1] = c2 * dr[1];          F.dk[2][2] = rfma(c2, dr[2], ir);
    }
}

// u̇^a = −Γ^a_bc u^b u^c for g = η + f k⊗k (k_t = 1, stationary).  Replaces christoffel + the contraction of
// geodesic (src/RayTraceGR.jl:321-331, :361-363) without forming Γ.  pos = (x,y,z), u = (u^t,u^x,u^y,u^z).
template <class R, bool FAST = false>
RTGR_DEV void ksform_accel(const KSField<R>& F, const R u[4], R ud[4]) {
    const R ut = u[0], ux = u[1], uy = u[2], uz = u[3];
    const R K = rfma(F.k[0], ux, rfma(F.k[1], uy, rfma(F.k[2], uz, ut)));          // k_a u^a
    const R Df = rfma(F.df[0], ux, rfma(F.df[1], uy, F.df[2] * uz));               // u·∇f
    R Dk[3], W[3];
    for (int i = 0; i < 3; i++) Dk[i] = rfma(F.dk[i][0], ux, rfma(F.dk[i][1], uy, F.dk[i][2] * uz));  // u^j ∂_j k_i
    for (int d = 0; d < 3; d++) W[d] = rfma(F.dk[0][d], ux, rfma(F.dk[1][d], uy, F.dk[2][d] * uz));   // u^i ∂_d k_i
    const R A = rfma(ux, Dk[0], rfma(uy, Dk[1], uz * Dk[2]));                      // u^b u^c ∂_b k_c
    const R P = rfma(K, Df, F.f * A);                                              // L_t
    const R fK = F.f * K, hK2 = R(-0.5) * K * K;
    R L[3];
    for (int i = 0; i < 3; i++) L[i] = rfma(F.k[i], P, rfma(fK, Dk[i] - W[i], hK2 * F.df[i]));
    const R kk = rfma(F.k[0], F.k[0], rfma(F.k[1], F.k[1], F.k[2] * F.k[2]));
    const R S = F.f * rcp_<FAST>(rfma(F.f, kk - R(1), R(1)));                      // f/(1 + f(|k|²−1))
    const R kL = rfma(F.k[0], L[0], rfma(F.k[1], L[1], rfma(F.k[2], L[2], -P)));   // k♯^d L_d, k♯ = (−1, k_i)
    const R SkL = S * kL;
    ud[0] = P - SkL;
    for (int i = 0; i < 3; i++) ud[1 + i] = rfma(F.k[i], SkL, -L[i]);
}

// Wave-uniform constants of the metric, formed ONCE per kernel (there is no scalar f64 ALU: a product of two kernel
// arguments is a VALU instruction and a VGPR pair unless it is pinned into SGPRs with readfirstlane, see uniform_()).
template <class R>
struct MetricK {
    R M, a;
    R M2;    // 2M
    R a2;    // a²
    R a2x2;  // 2a
};
// A 64-bit value every lane holds, made wave-uniform FOR THE COMPILER (SGPR pair).  __builtin_amdgcn_readfirstlane
// returns a signed int: each half goes through uint32_t, or a low word with bit 31 set sign-extends over the high word
// (round 2's spelling did exactly that — harmless for its operands, 100.0 and queue positions below 2^31, fatal for a² = 0.64).
RTGR_DEV unsigned long long uniform64(unsigned long long b) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
           (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
}
RTGR_DEV double uniform_(double v) {
    return __builtin_bit_cast(double, uniform64(__builtin_bit_cast(unsigned long long, v)));
}
RTGR_DEV float uniform_(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, v)));
}
template <class R>
RTGR_DEV MetricK<R> metric_consts(R M, R a) {
    MetricK<R> k;
    k.M = M; k.a = a;
    k.M2 = uniform_(R(2) * M);
    k.a2 = uniform_(a * a);
    k.a2x2 = uniform_(R(2) * a);
    return k;
}

// a = 0 (the reference's configuration, :276): k_i = x_i/r, every gradient is radial, and the whole contraction
// collapses to u̇^i = x_i·g, u̇^t = P − S·kL (derivation in DESIGN.md §RHS).  xs = (x,y,z), u = (u^t,u^x,u^y,u^z).
template <class R, int METRIC, bool FAST>
RTGR_DEV void accel_radial(const R xs[3], const R u[4], const MetricK<R>& C, R ud[4]) {
    const R rho2 = rfma(xs[0], xs[0], rfma(xs[1], xs[1], xs[2] * xs[2]));
    const R ir = FAST ? frsq<R>(rho2) : R(1) / rsqrt_(rho2);           // 1/ρ
    R invr, rq2;                                                         // 1/r ; ∇r = rq2·x
    if constexpr (METRIC == RTGR_KS_REF) {
        const R rho = rho2 * ir;                                         // r = (ρ + ρ²)/2  (:284 with a = 0)
        invr = R(2) * ir * rcp_<FAST>(R(1) + rho);
        rq2 = rfma(R(0.5), ir, R(1));
    } else {
        invr = ir;                                                       // r = ρ
        rq2 = ir;
    }
    const R f = C.M2 * invr;                                             // f = 2M r³/r⁴  (:285)
    const R xu = rfma(xs[0], u[1], rfma(xs[1], u[2], xs[2] * u[3]));
    const R uu = rfma(u[1], u[1], rfma(u[2], u[2], u[3] * u[3]));
    const R D = rq2 * xu;                                                // u·∇r
    const R Ku = invr * xu;                                              // k_i u^i
    const R K = u[0] + Ku;                                               // k_a u^a
    const R fi = f * invr;
    const R P = fi * rfma(-D, K + Ku, uu);                               // L_t = K Df + f A
    const R alpha = rfma(-fi * K, D, P);
    const R beta = fi * K * rfma(R(0.5), K, Ku);
    const R lam = rfma(invr, alpha, rq2 * beta);                         // L_i = x_i λ
    R S;
    if constexpr (METRIC == RTGR_KS_REF) {
        const R kk = rho2 * invr * invr;                                 // r ≠ ρ as written: |k|² ≠ 1
        S = f * rcp_<FAST>(rfma(f, kk - R(1), R(1)));
    } else {
        S = f;                                                           // |k|² = 1
    }
    const R kL = rfma(invr * lam, rho2, -P);                             // k♯^d L_d
    const R SkL = S * kL;
    ud[0] = P - SkL;
    const R g = rfma(SkL, invr, -lam);
    ud[1] = xs[0] * g; ud[2] = xs[1] * g; ud[3] = xs[2] * g;
}

// a != 0, the reference's as-written radius (:284): fused field + contraction.  ∂_j k_i = kr_i ∂_j r + E_ij with the sparse
// explicit part E = [[r w, a w, 0], [−a w, r w, 0], [0, 0, 1/r]] (w = 1/(r²+a²)), and ∇f = f_r ∇r + f_z ẑ, so the two
// contractions Dk_i = u^j ∂_j k_i and W_d = u^i ∂_d k_i come from D = u·∇r, kr·u, E u and Eᵀu without forming the 3x3
// Jacobian.  (Not a benchmark configuration: kept in the general form.)
template <class R, bool FAST>
RTGR_DEV void accel_spin_ref(const R xs[3], const R u[4], const MetricK<R>& C, R ud[4]) {
    const R x = xs[0], y = xs[1], z = xs[2];
    const R a = C.a, a2 = C.a2;
    const R rho2 = rfma(x, x, rfma(y, y, z * z));
    const R q = rho2 - a2;
    const R a2z = a2 * z;
    R s1, is1, s2, is2;                                   // r = sqrt(q)/2 + sqrt(a² z² + (q/2)²)            :284
    sqrt_inv<FAST>(q, s1, is1);
    const R hq = R(0.5) * q;
    sqrt_inv<FAST>(rfma(a2z, z, hq * hq), s2, is2);
    const R r = rfma(R(0.5), s1, s2);
    const R ir = rcp_<FAST>(r);
    const R rq2 = R(0.5) * rfma(q, is2, is1);             // ∇r = rq2 (x,y,z) + rz ẑ
    const R rz = a2z * is2;
    const R r2 = r * r;
    const R iden = rcp_<FAST>(rfma(r2, r2, a2z * z));     // 1/(r⁴ + a² z²)
    const R rid = r2 * r * iden;
    const R dr0 = rq2 * x, dr1 = rq2 * y, dr2 = rfma(rq2, z, rz);
    const R f = C.M2 * rid;                                     // f = 2M r³/(r⁴+a²z²)                             :285
    const R f_r = f * rfma(R(-4), rid, R(3) * ir);              // ∂f/∂r = f (3/r − 4r³/den)
    const R f_z = R(-2) * f * a2z * iden;                       // ∂f/∂z at fixed r
    const R w = rcp_<FAST>(r2 + a2);
    const R k0 = rfma(r, x, a * y) * w, k1 = rfma(r, y, -a * x) * w, k2 = z * ir;            // :286-289
    const R m2r = R(-2) * r;
    const R kr0 = w * rfma(m2r, k0, x), kr1 = w * rfma(m2r, k1, y), kr2 = -k2 * ir;          // ∂k_i/∂r
    const R rw = r * w, aw2 = C.a2x2 * w;
    const R ut = u[0], ux = u[1], uy = u[2], uz = u[3];
    const R D = rfma(dr0, ux, rfma(dr1, uy, dr2 * uz));        // u·∇r
    const R Df = rfma(f_r, D, f_z * uz);                        // u·∇f
    const R Ku = rfma(k0, ux, rfma(k1, uy, k2 * uz));
    const R K = ut + Ku;                                        // k_a u^a
    const R kru = rfma(kr0, ux, rfma(kr1, uy, kr2 * uz));
    // Dk_i = u^j ∂_j k_i = kr_i D + (E u)_i and W_d = u^i ∂_d k_i = kru ∂_d r + (Eᵀu)_d are needed only as
    //   A   = u·Dk      = kru D + uᵀE u          with uᵀE u = r w (ux² + uy²) + uz²/r   (the ±a w parts cancel)
    //   V_i = Dk_i − W_i = kr_i D − kru ∂_i r + ((E − Eᵀ) u)_i,   (E − Eᵀ) u = 2 a w (uy, −ux, 0)
    const R A = rfma(kru, D, rfma(rw, rfma(ux, ux, uy * uy), ir * uz * uz));
    const R V0 = rfma(kr0, D, rfma(-kru, dr0, aw2 * uy));
    const R V1 = rfma(kr1, D, rfma(-kru, dr1, -aw2 * ux));
    const R V2 = rfma(kr2, D, -kru * dr2);
    const R P = rfma(K, Df, f * A);                             // L_t
    const R hK2 = R(-0.5) * K * K;
    const R fK = f * K, g2 = hK2 * f_r, g3 = hK2 * f_z;
    const R L0 = rfma(k0, P, rfma(fK, V0, g2 * dr0));
    const R L1 = rfma(k1, P, rfma(fK, V1, g2 * dr1));
    const R L2 = rfma(k2, P, rfma(fK, V2, rfma(g2, dr2, g3)));
    const R kk = rfma(k0, k0, rfma(k1, k1, k2 * k2));
    const R S = f * rcp_<FAST>(rfma(f, kk - R(1), R(1)));
    const R kL = rfma(k0, L0, rfma(k1, L1, rfma(k2, L2, -P)));
    const R SkL = S * kL;
    ud[0] = P - SkL;
    ud[1] = rfma(k0, SkL, -L0);
    ud[2] = rfma(k1, SkL, -L1);
    ud[3] = rfma(k2, SkL, -L2);
}

// a != 0, TEXTBOOK radius — the configuration BASELINE.json words (a = 0.8, a = 0.998): 48 MUL + 41 FMA + 3 ADD + 2
// transcendental seeds per evaluation (round 2's form of the same contraction: 58 + 46 + 5 + 3).
//
// r is a root of r⁴ − q r² − a²z² = 0 (q = ρ² − a²) and k is the principal null congruence, which buys (checked to 40
// digits by tools/check_identities.py, which tests/test_identities.py runs): with Σ = sqrt(q² + 4a²z²) = 2r² − q,
//     r⁴ + a²z² = r² Σ;   |k|² = 1;   k^j ∂_j k_i = 0;   k^i ∂_d k_i = 0;   k·∇r = 1;
//     ∇r = (r/Σ)(x,y,z) + (a²z/(rΣ)) ẑ        — the SAME r/Σ that makes f = 2M r/Σ (one product, no reciprocal of its own);
//     ∂f/∂r = f ψ, ψ = 3/r − 4r/Σ;   ∂f/∂z|_r = f φ, φ = −2a²z/(r²Σ);   S = f/(1 + f(|k|²−1)) = f.
// With the general contraction written out (accel_spin_ref: A, V_i = Dk_i − W_i, L_i, P) and regrouped by WHAT MULTIPLIES
// x_i, k_i and the rotation (uy, −ux, 0) — neither ∂_i r, nor ∂k_i/∂r, nor V_i, nor L_i is ever formed —
//     u̇^t = f [ K (ψ D + φ u^z) + A ] + ½K² f² (ψ + φ k_z)
//     u̇^i = −k_i u̇^t − fK V_i + ½K² ∂_i f
//         = k_i (2 r w E − u̇^t) + x_i (c_r r/Σ − E w) − 2 a w fK (uy, −ux)_i                         (i = x, y)
//     u̇^z = z [ (E/r − u̇^t)/r + c_r r/Σ ] + c_r a²z/(rΣ) + ½K² f φ
// where D = u·∇r, K = k_a u^a, E = fK D, c_r = fK (∂k/∂r · u) + ½K² f ψ, A = u^b u^c ∂_b k_c.
template <class R, bool FAST>
RTGR_DEV void accel_spin_true(const R xs[3], const R u[4], const MetricK<R>& C, R ud[4]) {
    const R x = xs[0], y = xs[1], z = xs[2];
    const R ut = u[0], ux = u[1], uy = u[2], uz = u[3];
    const R q = rfma(x, x, rfma(y, y, rfma(z, z, -C.a2)));         // ρ² − a²
    const R a2z = C.a2 * z;
    const R sig2 = rfma(R(4), a2z * z, q * q);                      // Σ² = q² + 4a²z²
    R sig, is;                                                      // Σ, 1/Σ
    sqrt_inv<FAST>(sig2, sig, is);
    const R r2 = R(0.5) * (q + sig);                                // r² = (q + Σ)/2
    // 1/r AND w = 1/(r²+a²) from ONE reciprocal square root, t = 1/sqrt(r² (r²+a²)²) = 1/(r (r²+a²)):
    //     1/r = t (r²+a²),   r = r²/r,   w = t r
    // (an f64 transcendental seed costs 3.6 issue slots: one less per evaluation for three products more.  Float64 only:
    //  the chain t -> 1/r -> r -> w compounds three roundings, +10 % on the RHS's rounding noise, which Float32 — whose
    //  error estimate sits close to that noise, and whose seeds cost half — cannot afford)
    R r, ir, w;
    if constexpr (FAST && IsF64<R>::v) {
        const R r2a = r2 + C.a2;
        const R t = frsq<R>(r2 * (r2a * r2a));
        ir = t * r2a;
        r = r2 * ir;
        w = t * r;
    } else {
        sqrt_inv<FAST>(r2, r, ir);
        w = rcp_<FAST>(r2 + C.a2);
    }
    const R rid = r * is;                                           // r/Σ: ∇r's isotropic part AND r³/(r⁴+a²z²)
    const R rz = a2z * (is * ir);                                   // a²z/(rΣ): ∇r's extra z part
    const R phi = -(rz * ir) * R(2);                                // (∂f/∂z)/f
    const R psi = rfma(R(-4), rid, R(3) * ir);                      // (∂f/∂r)/f
    const R rw = r * w, aw = C.a * w;
    const R k0 = rfma(rw, x, aw * y), k1 = rfma(rw, y, -(aw * x)), k2 = z * ir;                 // :286-289
    const R xu2 = rfma(x, ux, y * uy), xu = rfma(z, uz, xu2);
    const R Ku2 = rfma(k0, ux, k1 * uy);
    const R K = ut + rfma(k2, uz, Ku2);                             // k_a u^a
    const R D = rfma(rid, xu, rz * uz);                             // u·∇r
    const R rw2m = R(-2) * rw;
    const R iruz = ir * uz;
    // ∂k/∂r · u with ∂k_i/∂r = w (x_i − 2 r k_i) (i = x, y), −k_z/r (z)
    const R kru = rfma(w, xu2, rfma(rw2m, Ku2, -(k2 * iruz)));
    // A = u^b u^c ∂_b k_c = kru D + uᵀE u,  uᵀE u = r w (ux² + uy²) + uz²/r  (the ±a w parts of E cancel)
    const R A = rfma(kru, D, rfma(rw, rfma(ux, ux, uy * uy), iruz * uz));
    const R inner = rfma(psi, D, phi * uz);                         // (u·∇f)/f
    const R f = C.M2 * rid;                                         // f = 2M r³/(r⁴+a²z²) = 2M r/Σ                  :285
    const R Kf = f * K, hf = (R(0.5) * K) * Kf;                     // fK, ½K² f
    const R udt = rfma(hf * f, rfma(phi, k2, psi), rfma(Kf, inner, f * A));
    const R E = Kf * D;
    const R cr = rfma(Kf, kru, hf * psi);
    const R nck = rfma(rw2m, E, udt);                               // −(2 r w E − u̇^t)
    const R crr = cr * rid;
    const R cx = rfma(-E, w, crr);
    const R rot = (C.a2x2 * w) * Kf;                                // 2 a w fK
    ud[0] = udt;
    ud[1] = rfma(-k0, nck, rfma(x, cx, -(rot * uy)));
    ud[2] = rfma(-k1, nck, rfma(y, cx, rot * ux));
    ud[3] = rfma(z, rfma(ir, rfma(E, ir, -udt), crr), rfma(cr, rz, hf * phi));
}

// acceleration only (the ẋ = u half is handled by the caller):  u̇ = accel(x_spatial, u)
template <class R> RTGR_DEV void accel_generic(uint32_t metric, R xt, const R xs[3], const R u[4], R M, R a, R ud[4]);
constexpr int RTGR_GENERIC_BASE = 100;  // METRIC template value 100 + kind selects the generic dual-number RHS
#ifndef RTGR_USER_NE
#define RTGR_USER_NE 4   // a user unit built with -DRTGR_USER_NE=3 declares its metric stationary (api.UserMetric(stationary=True))
#endif

// Does the integrate loop have to carry the stage's TIME coordinate for this metric?  The reference evaluates
// christoffel(metric, x) at the full 4-position (src/RayTraceGR.jl:358-363); every built-in metric is stationary, so only
// a user metric that was NOT declared stationary needs x^t at the stages (the other instantiations never form it).
template <int METRIC> constexpr bool needs_stage_time() {
#ifdef RTGR_USER_METRIC
    return METRIC == RTGR_GENERIC_BASE + RTGR_USER && RTGR_USER_NE == 4;
#else
    return false;
#endif
}

// xt: the time coordinate of the evaluation point (read only when needs_stage_time<METRIC>())
template <class R, int METRIC, bool SPIN, bool FAST>
RTGR_DEV void accel(const R xs[3], const R u[4], const MetricK<R>& C, R ud[4], R xt = R(0)) {
    if constexpr (METRIC >= RTGR_GENERIC_BASE) {
        accel_generic<R>((uint32_t)(METRIC - RTGR_GENERIC_BASE), xt, xs, u, C.M, C.a, ud);
    } else if constexpr (METRIC == RTGR_MINKOWSKI) {
        ud[0] = ud[1] = ud[2] = ud[3] = R(0);
    } else if constexpr (!SPIN) {
        accel_radial<R, METRIC, FAST>(xs, u, C, ud);
    } else if constexpr (METRIC == RTGR_KS_REF) {
        accel_spin_ref<R, FAST>(xs, u, C, ud);
    } else {
        accel_spin_true<R, FAST>(xs, u, C, ud);
    }
}

// geodesic RHS, production path.  s = (x^a, u^a) -> (u^a, u̇^a)                       src/RayTraceGR.jl:358-370
template <class R, int METRIC, bool SPIN>
RTGR_DEV void rhs(const R s[8], R M, R a, R ds[8]) {
    ds[0] = s[4]; ds[1] = s[5]; ds[2] = s[6]; ds[3] = s[7];
    if constexpr (METRIC == RTGR_MINKOWSKI) {
        ds[4] = ds[5] = ds[6] = ds[7] = R(0);  // Γ ≡ 0
    } else {
        KSField<R> F;
        ks_field<R, METRIC, SPIN>(s[1], s[2], s[3], M, a, F);
        ksform_accel<R>(F, s + 4, ds + 4);
    }
}

// ---- generic forward-mode path ----------------------------------------------------------------------------------------
// Dual{T,DT} (src/RayTraceGR.jl:11-14) on the device: value + NE partials, in registers.
//   NE   = 4: DT = SVector{4,T}, the reference's choice (all four coordinates seeded, :303-306)
//   NE   = 3: the spatial partials only — for a STATIONARY metric ∂_t g ≡ 0, so the t-seed carries exact zeros through
//          every operation (a quarter of the dual arithmetic); the integrate kernels use it for the built-in metrics and
//          for user metrics declared stationary
//   FAST = true: the reciprocals and square roots inside `/` and `sqrt` are the 4 / 6-instruction sequences of the hot
//          loop (frcp / frsq, <= 1.5e-16 relative) instead of the 11 / 14-instruction IEEE expansions
template <class R, int NE = 4, bool FAST = false>
struct DDual {
    R v, e[NE];
};
#define RTGR_DT template <class R, int NE, bool FAST>
#define RTGR_DD DDual<R, NE, FAST>
// a plain-scalar operand of a mixed operation takes the dual's scalar type without taking part in template deduction, so
// that `0.5 * q` works on Float32 duals too (the literal is a double)
template <class T> struct NoDeduce { using type = T; };
#define RTGR_SC typename NoDeduce<R>::type
RTGR_DT RTGR_DEV RTGR_DD dconst_(R v) { RTGR_DD r; r.v = v; for (int i = 0; i < NE; i++) r.e[i] = R(0); return r; }
template <class R> RTGR_DEV DDual<R> dconst(R v) { return dconst_<R, 4, false>(v); }
RTGR_DT RTGR_DEV RTGR_DD operator+(const RTGR_DD& x, const RTGR_DD& y) {
    RTGR_DD r; r.v = x.v + y.v; for (int i = 0; i < NE; i++) r.e[i] = x.e[i] + y.e[i]; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator-(const RTGR_DD& x, const RTGR_DD& y) {
    RTGR_DD r; r.v = x.v - y.v; for (int i = 0; i < NE; i++) r.e[i] = x.e[i] - y.e[i]; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator+(const RTGR_DD& x, RTGR_SC a) { RTGR_DD r = x; r.v = x.v + a; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator-(const RTGR_DD& x, RTGR_SC a) { RTGR_DD r = x; r.v = x.v - a; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator*(const RTGR_DD& x, const RTGR_DD& y) {
    RTGR_DD r; r.v = x.v * y.v; for (int i = 0; i < NE; i++) r.e[i] = rfma(x.e[i], y.v, x.v * y.e[i]); return r; }
RTGR_DT RTGR_DEV RTGR_DD operator*(RTGR_SC a, const RTGR_DD& x) {
    RTGR_DD r; r.v = a * x.v; for (int i = 0; i < NE; i++) r.e[i] = a * x.e[i]; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator*(const RTGR_DD& x, RTGR_SC a) { return a * x; }
RTGR_DT RTGR_DEV RTGR_DD operator/(const RTGR_DD& x, const RTGR_DD& y) {
    const R iy = rcp_<FAST, R>(y.v); RTGR_DD r; r.v = x.v * iy;
    for (int i = 0; i < NE; i++) r.e[i] = rfma(-r.v, y.e[i], x.e[i]) * iy; return r; }
RTGR_DT RTGR_DEV RTGR_DD dsqrt(const RTGR_DD& x) {
    RTGR_DD r; R is; sqrt_inv<FAST, R>(x.v, r.v, is); const R c = R(0.5) * is;
    for (int i = 0; i < NE; i++) r.e[i] = c * x.e[i]; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator-(const RTGR_DD& x) {
    RTGR_DD r; r.v = -x.v; for (int i = 0; i < NE; i++) r.e[i] = -x.e[i]; return r; }
RTGR_DT RTGR_DEV RTGR_DD operator+(RTGR_SC a, const RTGR_DD& x) { return x + a; }
RTGR_DT RTGR_DEV RTGR_DD operator-(RTGR_SC a, const RTGR_DD& x) { return (-x) + a; }
RTGR_DT RTGR_DEV RTGR_DD operator/(const RTGR_DD& x, RTGR_SC a) { return rcp_<FAST, R>(a) * x; }
RTGR_DT RTGR_DEV RTGR_DD operator/(RTGR_SC a, const RTGR_DD& x) { return dconst_<R, NE, FAST>(a) / x; }
// f(x) with derivative f'(x): the chain rule every elementary function below is an instance of
RTGR_DT RTGR_DEV RTGR_DD dchain(const RTGR_DD& x, R fv, R dfv) {
    RTGR_DD r; r.v = fv; for (int i = 0; i < NE; i++) r.e[i] = dfv * x.e[i]; return r; }

// ---- user metrics (RTGR_USER): "the metric is any callable" of the reference (src/RayTraceGR.jl:302-309) -------------
// A translation unit that defines RTGR_USER_METRIC supplies
//     template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]);
// written with + - * / and the m* helpers below, so that it runs on plain scalars AND on forward duals — exactly the
// contract the reference puts on a metric function.  Such units are generated, compiled with hipcc --genco and loaded
// at run time (rtgr_user_metric_load, api.UserMetric); see rtgr_user_unit.hip.in.
// The helpers are the elementary functions the reference's Dual carries (src/RayTraceGR.jl:132-196):
//     msqrt mexp mlog msin mcos (:171-196)   mabs (:150)   macos masin matan (:154-163)   matan2(y, x) (:165-169)
//     mcbrt (:171)   mpow(x, p) with a real exponent (:138-148; integer powers are better written as products)
// with the textbook derivatives.  (The reference's own atan(y,x) drops a 1/ρ² on its first term, `\` has a sign flipped
// and `scalar - Dual` keeps +eps — SURVEY §4.3: off the reference's hot path, not reproduced here; the oracle's twins of
// these helpers are the textbook forms too, see oracle/rtgr_oracle.cpp.)
template <class R> RTGR_DEV R mfun_sqrt(R x) { return rsqrt_(x); }
RTGR_DEV double msqrt(double x) { return __builtin_sqrt(x); }
RTGR_DEV float msqrt(float x) { return __builtin_sqrtf(x); }
RTGR_DT RTGR_DEV RTGR_DD msqrt(const RTGR_DD& x) { return dsqrt(x); }
RTGR_DEV double mexp(double x) { return exp(x); }
RTGR_DEV float mexp(float x) { return expf(x); }
RTGR_DT RTGR_DEV RTGR_DD mexp(const RTGR_DD& x) { const R f = mexp(x.v); return dchain(x, f, f); }
RTGR_DEV double mlog(double x) { return log(x); }
RTGR_DEV float mlog(float x) { return logf(x); }
RTGR_DT RTGR_DEV RTGR_DD mlog(const RTGR_DD& x) { return dchain(x, mlog(x.v), R(1) / x.v); }
RTGR_DEV double msin(double x) { return sin(x); }
RTGR_DEV float msin(float x) { return sinf(x); }
RTGR_DEV double mcos(double x) { return cos(x); }
RTGR_DEV float mcos(float x) { return cosf(x); }
RTGR_DT RTGR_DEV RTGR_DD msin(const RTGR_DD& x) { return dchain(x, msin(x.v), mcos(x.v)); }
RTGR_DT RTGR_DEV RTGR_DD mcos(const RTGR_DD& x) { return dchain(x, mcos(x.v), -msin(x.v)); }
RTGR_DEV double mabs(double x) { return __builtin_fabs(x); }
RTGR_DEV float mabs(float x) { return __builtin_fabsf(x); }
RTGR_DT RTGR_DEV RTGR_DD mabs(const RTGR_DD& x) { return dchain(x, rabs(x.v), x.v < R(0) ? R(-1) : R(1)); }
RTGR_DEV double macos(double x) { return acos(x); }
RTGR_DEV float macos(float x) { return acosf(x); }
RTGR_DT RTGR_DEV RTGR_DD macos(const RTGR_DD& x) { return dchain(x, macos(x.v), R(-1) / rsqrt_(R(1) - x.v * x.v)); }
RTGR_DEV double masin(double x) { return asin(x); }
RTGR_DEV float masin(float x) { return asinf(x); }
RTGR_DT RTGR_DEV RTGR_DD masin(const RTGR_DD& x) { return dchain(x, masin(x.v), R(1) / rsqrt_(R(1) - x.v * x.v)); }
RTGR_DEV double matan(double x) { return atan(x); }
RTGR_DEV float matan(float x) { return atanf(x); }
RTGR_DT RTGR_DEV RTGR_DD matan(const RTGR_DD& x) { return dchain(x, matan(x.v), R(1) / (R(1) + x.v * x.v)); }
RTGR_DEV double matan2(double y, double x) { return atan2(y, x); }
RTGR_DEV float matan2(float y, float x) { return atan2f(y, x); }
RTGR_DT RTGR_DEV RTGR_DD matan2(const RTGR_DD& y, const RTGR_DD& x) {   // d atan2 = (x dy − y dx)/(x² + y²)
    const R ir2 = R(1) / (x.v * x.v + y.v * y.v);
    RTGR_DD r; r.v = matan2(y.v, x.v);
    for (int i = 0; i < NE; i++) r.e[i] = (x.v * y.e[i] - y.v * x.e[i]) * ir2;
    return r; }
RTGR_DEV double mcbrt(double x) { return cbrt(x); }
RTGR_DEV float mcbrt(float x) { return cbrtf(x); }
RTGR_DT RTGR_DEV RTGR_DD mcbrt(const RTGR_DD& x) { const R c = mcbrt(x.v); return dchain(x, c, R(1) / (R(3) * c * c)); }
RTGR_DEV double mpow(double x, double p) { return pow(x, p); }
RTGR_DEV float mpow(float x, double p) { return powf(x, (float)p); }
RTGR_DT RTGR_DEV RTGR_DD mpow(const RTGR_DD& x, double p) {           // x^p, real constant exponent: p x^(p-1)
    const R f = mpow(x.v, (R)p); return dchain(x, f, (R)p * f / x.v); }
template <class S> struct MConst;                                     // a constant of the scalar type S
template <> struct MConst<double> { static RTGR_DEV double make(double c) { return c; } };
template <> struct MConst<float> { static RTGR_DEV float make(double c) { return (float)c; } };
RTGR_DT struct MConst<RTGR_DD> { static RTGR_DEV RTGR_DD make(double c) { return dconst_<R, NE, FAST>((R)c); } };
template <class S> RTGR_DEV S mconst(double c) { return MConst<S>::make(c); }
#ifdef RTGR_USER_METRIC
template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]);
#ifdef RTGR_USER_KS
// A user metric OF KERR–SCHILD FORM, g = η + f k⊗k with k_t = 1, k null with respect to η (|k⃗|² = 1: what makes it a
// Kerr–Schild metric and g⁻¹ = η⁻¹ − f k♯k♯) and no t-dependence, may be given by its two ingredients instead of its 16
// entries: the unit then defines rtgr_user_ks (and the unit template derives rtgr_user_metric from it for the camera and the
// evaluation hooks).  The integrate kernels push 3-wide duals through FOUR scalars (f, k_x, k_y, k_z) instead of ten
// metric entries and contract with ksform_accel — no 4x4 solve, no per-direction matrix–vector products.
template <class S> __device__ void rtgr_user_ks(const S x[4], double M, double a, S& f, S k[3]);
#endif
#endif

// metric(x::SVector{4,Dual}) for the built-ins; the UPPER triangle g[a][b], a <= b, is filled (and mirrored).
// Written the way one would write kerr_schild for speed: k_t = 1 is a constant, so g_tt = −1 + f, g_ti = f k_i and only
// g_ij = δ_ij + (f k_i) k_j costs dual products (9 of them instead of the 20 of `f * k[a] * k[b]` over all a <= b, :291).
RTGR_DT RTGR_DEV void metric_dual(uint32_t metric, R M, R a, const RTGR_DD xx[4], RTGR_DD g[4][4]) {
#ifdef RTGR_USER_METRIC
    if (metric == (uint32_t)RTGR_USER) {
        rtgr_user_metric<RTGR_DD>(xx, (double)M, (double)a, g);
        return;
    }
#endif
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < 4; q++) g[p][q] = dconst_<R, NE, FAST>(p == q ? (p == 0 ? R(-1) : R(1)) : R(0));   // η  :263,:282
    if (metric == RTGR_MINKOWSKI) return;
    const RTGR_DD &x = xx[1], &y = xx[2], &z = xx[3];
    const R a2 = a * a;
    const RTGR_DD zz = z * z;
    RTGR_DD rho2 = x * x + y * y + zz;                                                              // :283
    RTGR_DD r;
    if (metric == RTGR_KS_REF) {
        RTGR_DD q = rho2 - a2;
        RTGR_DD hq = R(0.5) * q;
        r = R(0.5) * dsqrt(q) + dsqrt(a2 * zz + hq * hq);                                           // :284
    } else {
        RTGR_DD q = rho2 - a2;
        r = dsqrt(R(0.5) * (q + dsqrt(q * q + (R(4) * a2) * zz)));
    }
    RTGR_DD r2 = r * r;
    RTGR_DD f = ((R(2) * M) * (r2 * r)) / (r2 * r2 + a2 * zz);                                      // :285
    RTGR_DD k[4];
    RTGR_DD den = r2 + a2;                                                                          // :286-289
    k[1] = (r * x + a * y) / den;
    k[2] = (r * y - a * x) / den;
    k[3] = z / r;
    g[0][0] = f - R(1);                                                                             // η_tt + f k_t k_t   :291
    for (int p = 1; p < 4; p++) {
        const RTGR_DD fk = f * k[p];
        g[0][p] = fk;                                                                               // f k_t k_p
        for (int q = p; q < 4; q++) g[p][q] = g[p][q] + fk * k[q];
    }
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < p; q++) g[p][q] = g[q][p];
}

template <class R>
RTGR_DEV void inv4sym(const R m[4][4], R o[4][4]) {  // cofactor inverse (StaticArrays closed form)
#pragma clang fp contract(off)   // (part of the camera: see make_pixel, rtgr_integrator.hpp)
    const R s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1], s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    const R s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3], s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    const R s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3], s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    const R c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const R c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const R c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const R id = R(1) / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
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

// x = m^{-1} b for a SYMMETRIC 4x4 m (only m[i][j], i <= j, is read): adjugate by 2x2 minors (robust where g_tt = 0, which
// elimination without pivoting is not), 10 cofactors instead of 16, one reciprocal.  SURVEY §8d "symmetric inverse by
// 2x2 minors".
template <class R, bool FAST>
RTGR_DEV void solve4sym(const R m[4][4], const R b[4], R x[4]) {
    const R m00 = m[0][0], m01 = m[0][1], m02 = m[0][2], m03 = m[0][3], m11 = m[1][1], m12 = m[1][2], m13 = m[1][3],
            m22 = m[2][2], m23 = m[2][3], m33 = m[3][3];
    // minors of rows {0,1} (s) and rows {2,3} (c), columns (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    const R s0 = rfma(m00, m11, -m01 * m01), s1 = rfma(m00, m12, -m01 * m02), s2 = rfma(m00, m13, -m01 * m03);
    const R s3 = rfma(m01, m12, -m11 * m02), s4 = rfma(m01, m13, -m11 * m03), s5 = rfma(m02, m13, -m12 * m03);
    const R c0 = rfma(m02, m13, -m03 * m12), c1 = rfma(m02, m23, -m03 * m22), c2 = rfma(m02, m33, -m03 * m23);
    const R c3 = rfma(m12, m23, -m13 * m22), c4 = rfma(m12, m33, -m13 * m23), c5 = rfma(m22, m33, -m23 * m23);
    const R det = rfma(s0, c5, rfma(-s1, c4, rfma(s2, c3, rfma(s3, c2, rfma(-s4, c1, s5 * c0)))));
    const R id = rcp_<FAST, R>(det);
    // adjugate (symmetric): A_ij = cofactor_ji
    const R a00 = rfma(m11, c5, rfma(-m12, c4, m13 * c3));
    const R a01 = -rfma(m01, c5, rfma(-m02, c4, m03 * c3));
    const R a02 = rfma(m13, s5, rfma(-m23, s4, m33 * s3));
    const R a03 = -rfma(m12, s5, rfma(-m22, s4, m23 * s3));
    const R a11 = rfma(m00, c5, rfma(-m02, c2, m03 * c1));
    const R a12 = -rfma(m03, s5, rfma(-m23, s2, m33 * s1));
    const R a13 = rfma(m02, s5, rfma(-m22, s2, m23 * s1));
    const R a22 = rfma(m03, s4, rfma(-m13, s2, m33 * s0));
    const R a23 = -rfma(m02, s4, rfma(-m12, s2, m23 * s0));
    const R a33 = rfma(m02, s3, rfma(-m12, s1, m22 * s0));
    x[0] = id * rfma(a00, b[0], rfma(a01, b[1], rfma(a02, b[2], a03 * b[3])));
    x[1] = id * rfma(a01, b[0], rfma(a11, b[1], rfma(a12, b[2], a13 * b[3])));
    x[2] = id * rfma(a02, b[0], rfma(a12, b[1], rfma(a22, b[2], a23 * b[3])));
    x[3] = id * rfma(a03, b[0], rfma(a13, b[1], rfma(a23, b[2], a33 * b[3])));
}

// dmetric (src/RayTraceGR.jl:302-313): g[a][b], dg[a][b][c] = ∂_c g_ab
// UPPER: read only the upper triangle of what the metric function returned (a metric is symmetric; the reference
// asserts it, :307).  rtgr_eval_metric_f64 reports the function's output as is (UPPER = false), which is where an
// asymmetric user metric shows.
template <class R, bool UPPER = false>
RTGR_DEV void dmetric_dev(uint32_t metric, R M, R a, const R x[4], R g[4][4], R dg[4][4][4]) {
    DDual<R> xdx[4], gd[4][4];
    for (int p = 0; p < 4; p++) {
        xdx[p] = dconst<R>(x[p]);
        xdx[p].e[p] = R(1);
    }
    metric_dual<R, 4, false>(metric, M, a, xdx, gd);
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < 4; q++) {
            const DDual<R>& e = (UPPER && q < p) ? gd[q][p] : gd[p][q];
            g[p][q] = e.v;
            for (int c = 0; c < 4; c++) dg[p][q][c] = e.e[c];
        }
}

// The generic acceleration u̇ = −g^{-1} L,  L_d = ∂_b g_dc u^b u^c − ½ ∂_d g_bc u^b u^c  (christoffel + the contraction of
// geodesic, src/RayTraceGR.jl:321-331, :361-363, contracted BEFORE raising — SURVEY §8d's lean form).
// Per derivative direction j the symmetric matrix G_j = ∂_j g is applied to u once: v_j = G_j u (16 FMA); then
//     ∂_b g_dc u^b u^c = Σ_j u^j (v_j)_d        and        ∂_d g_bc u^b u^c = u · v_d
// — 20 FMA per direction instead of the 2 x 16 x 2 of the two double sums over (b, c).  Only the upper triangle of the
// metric function's output is read.  NE = 3: directions x, y, z only (stationary metric: ∂_t g = 0 exactly).
template <class R, int NE, bool FAST>
RTGR_DEV void generic_accel(uint32_t metric, R M, R a, const R x[4], const R u[4], R ud[4]) {
    constexpr int J0 = 4 - NE;  // first seeded coordinate
    DDual<R, NE, FAST> xdx[4], gd[4][4];
    for (int p = 0; p < 4; p++) {
        xdx[p] = dconst_<R, NE, FAST>(x[p]);
        if (p >= J0) xdx[p].e[p - J0] = R(1);
    }
    metric_dual<R, NE, FAST>(metric, M, a, xdx, gd);
    R t1[4] = {R(0), R(0), R(0), R(0)}, t2[NE];
#pragma unroll
    for (int j = 0; j < NE; j++) {
        R v[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            R acc = R(0);
#pragma unroll
            for (int c = 0; c < 4; c++) acc = rfma((d <= c ? gd[d][c] : gd[c][d]).e[j], u[c], acc);
            v[d] = acc;
        }
        t2[j] = rfma(u[0], v[0], rfma(u[1], v[1], rfma(u[2], v[2], u[3] * v[3])));
#pragma unroll
        for (int d = 0; d < 4; d++) t1[d] = rfma(u[J0 + j], v[d], t1[d]);
    }
    R L[4], g[4][4];
#pragma unroll
    for (int d = 0; d < 4; d++) L[d] = d >= J0 ? rfma(R(0.5), t2[d - J0], -t1[d]) : -t1[d];   // −L
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = p; q < 4; q++) g[p][q] = gd[p][q].v;
    solve4sym<R, FAST>(g, L, ud);
}

// generic RHS as a function of the 8-vector state (parity hook rtgr_eval_geodesic path 1: IEEE division, all four
// coordinates seeded — the reference's formulation to the letter)
template <class R>
RTGR_DEV void generic_rhs(uint32_t metric, R M, R a, const R s[8], R ds[8]) {
    for (int p = 0; p < 4; p++) ds[p] = s[4 + p];
    generic_accel<R, 4, false>(metric, M, a, s, s + 4, ds + 4);
}

// acceleration through the GENERIC dual-number path (what the reference does for any metric callable): used by the
// integrate kernels when the scene asks for it (RTGR_METRIC_GENERIC flag) and for every user metric.
#if defined(RTGR_USER_METRIC) && defined(RTGR_USER_KS)
// u̇ for a user metric given in Kerr–Schild form (rtgr_user_ks): (f, ∇f, k_i, ∂_j k_i) from 3-wide forward duals through the
// user's function, then the closed contraction of the built-ins' IEEE path with S = f (k null).
template <class R>
RTGR_DEV void user_ks_accel(const R x[4], const R u[4], R M, R a, R ud[4]) {
    using Dd = DDual<R, 3, true>;
    Dd xd[4], f, k[3];
    for (int p = 0; p < 4; p++) {
        xd[p] = dconst_<R, 3, true>(x[p]);
        if (p >= 1) xd[p].e[p - 1] = R(1);
    }
    rtgr_user_ks<Dd>(xd, (double)M, (double)a, f, k);
    const R ut = u[0], ux = u[1], uy = u[2], uz = u[3];
    const R K = rfma(k[0].v, ux, rfma(k[1].v, uy, rfma(k[2].v, uz, ut)));                       // k_a u^a
    const R Df = rfma(f.e[0], ux, rfma(f.e[1], uy, f.e[2] * uz));                               // u·∇f
    R Dk[3], W[3];
    for (int i = 0; i < 3; i++) Dk[i] = rfma(k[i].e[0], ux, rfma(k[i].e[1], uy, k[i].e[2] * uz));        // u^j ∂_j k_i
    for (int d = 0; d < 3; d++) W[d] = rfma(k[0].e[d], ux, rfma(k[1].e[d], uy, k[2].e[d] * uz));         // u^i ∂_d k_i
    const R A = rfma(ux, Dk[0], rfma(uy, Dk[1], uz * Dk[2]));                                   // u^b u^c ∂_b k_c
    const R P = rfma(K, Df, f.v * A);                                                           // L_t
    const R fK = f.v * K, hK2 = R(-0.5) * K * K;
    R L[3];
    for (int i = 0; i < 3; i++) L[i] = rfma(k[i].v, P, rfma(fK, Dk[i] - W[i], hK2 * f.e[i]));
    const R kL = rfma(k[0].v, L[0], rfma(k[1].v, L[1], rfma(k[2].v, L[2], -P)));                // k♯^d L_d, k♯ = (−1, k_i)
    const R SkL = f.v * kL;                                                                     // S = f/(1 + f(|k|²−1)) = f
    ud[0] = P - SkL;
    for (int i = 0; i < 3; i++) ud[1 + i] = rfma(k[i].v, SkL, -L[i]);
}
#endif

template <class R>
RTGR_DEV void accel_generic(uint32_t metric, R xt, const R xs[3], const R u[4], R M, R a, R ud[4]) {
    const R x[4] = {xt, xs[0], xs[1], xs[2]};   // (xt = 0 for every stationary metric: it is never read)
#ifdef RTGR_USER_METRIC
#ifdef RTGR_USER_KS
    if (metric == (uint32_t)RTGR_USER) { user_ks_accel<R>(x, u, M, a, ud); return; }
#endif
    if (metric == (uint32_t)RTGR_USER) { generic_accel<R, RTGR_USER_NE, true>(metric, M, a, x, u, ud); return; }
#endif
    generic_accel<R, 3, true>(metric, M, a, x, u, ud);   // the built-in metrics are stationary
}

// christoffel (src/RayTraceGR.jl:321-331): all 64 entries, for rtgr_eval_metric_f64
template <class R>
RTGR_DEV void christoffel_dev(const R g[4][4], const R dg[4][4][4], R Gam[4][4][4]) {
    R gu[4][4];
    inv4sym<R>(g, gu);
    for (int p = 0; p < 4; p++)
        for (int b = 0; b < 4; b++)
            for (int c = 0; c < 4; c++) {
                R acc = R(0);
                for (int d = 0; d < 4; d++) acc = rfma(gu[p][d], R(0.5) * (dg[d][b][c] + dg[d][c][b] - dg[b][c][d]), acc);
                Gam[p][b][c] = acc;
            }
}

// ---- objects (src/RayTraceGR.jl:374-441) -------------------------------------------------------------------------------
// `Object{T}` is an open abstract type with two methods, distance and objcolor (:374-389).  A run-time unit whose source
// defines them (rtgr_user_unit.hip.in sets RTGR_USER_OBJECTS) supplies
//     template <class S> __device__ S    rtgr_user_distance(unsigned type, const S x[4], const S p[9]);
//     template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]);
// and, optionally (RTGR_USER_REACH), the bound the FAR pass needs to skip a step's scan:
//     template <class S> __device__ S    rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]);
//     >= |distance(x') − distance(x)| for every x' with |x'_q − x_q| <= dl[q]
// (include/rtgr.h "user objects").  The library's own kernels are compiled without them: a scene with an RTGR_USER_OBJECT
// only ever runs with the kernels of its unit (convert_scene, rtgr_context.hip).
#ifdef RTGR_USER_OBJECTS
template <class S> __device__ S rtgr_user_distance(unsigned type, const S x[4], const S p[9]);
template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]);
#ifdef RTGR_USER_REACH
template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]);
#endif
#ifdef RTGR_USER_SAMPLE
template <class S> __device__ bool rtgr_user_sample(unsigned type, S p[9]);   // optional: a sample object of `type` for the load-time probe
#endif
#endif

template <class R>
RTGR_DEV R obj_distance(const DevObject<R>& o, const R pos[4]) {
    if (o.kind == RTGR_PLANE) return pos[0] - o.p[0];                                    // :399-401
    if (o.kind == RTGR_SPHERE) {                                                         // :415-419
        const R dx = pos[1] - o.p[1], dy = pos[2] - o.p[2], dz = pos[3] - o.p[3];
        const R Rr = o.p[8];
        const R d = rfma(dx, dx, rfma(dy, dy, rfma(dz, dz, -Rr * Rr)));
        return Rr < R(0) ? -d : d;  // sign(R)*( … ); R = 0 never used
    }
#ifdef RTGR_USER_OBJECTS
    if (o.kind == RTGR_USER_OBJECT) return rtgr_user_distance<R>(o.type, pos, o.p);     // distance(obj::MyThing, pos)  :377-386
#endif
    // RTGR_DISK: max(|z|−h, r_in−ϱ, ϱ−r_out)
    const R rc = rsqrt_(rfma(pos[1], pos[1], pos[2] * pos[2]));
    R d = rabs(pos[3]) - o.p[0];
    d = rmax(d, o.p[1] - rc);
    d = rmax(d, rc - o.p[2]);
    return d;
}

// The disk's distance as the ContinuousCallback SCAN needs it: its sign only (the scan multiplies the minimum over the
// objects by the sign at the step start and tests < 0 / <= 0; the minimum's sign is fixed by its members' signs).  The two
// radial terms r_in − ϱ and ϱ − r_out are replaced by THEIR SIGNS, read off s = x² + y² without taking the root: the host
// precomputes, in the device's scalar type, the band of s whose correctly rounded square root equals the radius
// (p[3] = min{s : √s >= r_in}, p[4] = min{s : √s > r_in}, p[5], p[6] likewise for r_out; disk_sqrt_band, rtgr_context.hip), so
//     sign(r_in − RN(√s)) = +1 for s < p[3], 0 for p[3] <= s < p[4], −1 otherwise
// EXACTLY — same sign, zero included, as obj_distance computes with its IEEE square root, for every s (√ is monotone and
// correctly rounded).  Nine IEEE roots (~16 instructions each) per accepted NEAR step become compares and selects; the true
// distance is still what the event root-finder (resolve_kernel) and the colouring see.
template <class R>
RTGR_DEV R disk_sign_distance(const DevObject<R>& o, R px, R py, R pz) {
    const R s = rfma(px, px, py * py);
    const R e_in = s < o.p[3] ? R(1) : (s < o.p[4] ? R(0) : R(-1));     // sign(r_in − ϱ)
    const R e_out = s < o.p[5] ? R(-1) : (s < o.p[6] ? R(0) : R(1));    // sign(ϱ − r_out)
    return rmax(rmax(rabs(pz) - o.p[0], e_in), e_out);
}

// … and as the FAR pass's reach bound needs it: its magnitude, to ~1e-16 relative (the bound carries a 1e-6 guard), from
// the 6-instruction reciprocal square root instead of the IEEE expansion.
template <class R>
RTGR_DEV R disk_distance_fast(const DevObject<R>& o, R px, R py, R pz) {
    const R s = rfma(px, px, py * py);
    const R rc = s > R(0) ? s * frsq<R>(s) : R(0);
    return rmax(rmax(rabs(pz) - o.p[0], o.p[1] - rc), rc - o.p[2]);
}

// The object list in order — f(object, index) —, as `for obj in objs` walks the reference's Vector (:434, :520): the first
// RTGR_MAX_OBJECTS objects from the kernels' argument block (a wave-uniform index into the kernarg segment: scalar loads), the
// rest of a longer list from the scene's device table (DevScene::more; the same wave-uniform walk over global memory).  The second
// loop is cold code for every scene of up to RTGR_MAX_OBJECTS objects: never entered, and outside the hot loop's instruction
// stream.
template <class R, class F>
RTGR_DEV void for_each_object(const DevScene<R>& sc, F&& f) {
    const uint32_t n0 = sc.nobj < (uint32_t)RTGR_MAX_OBJECTS ? sc.nobj : (uint32_t)RTGR_MAX_OBJECTS;
    for (uint32_t o = 0; o < n0; o++) f(sc.obj[o], o);
#ifndef RTGR_INLINE_OBJECTS_ONLY   // (A/B builds: the loop as it was before lists could be longer — tools/launch_ab.py builds)
    if (__builtin_expect(sc.nobj > (uint32_t)RTGR_MAX_OBJECTS, 0)) {   // (laid out of line: 0.4-0.7 % of the 4096² frame when it sat in the hot loop's stream)
        // The table is read-only for the kernel's lifetime and walked with a wave-uniform index: through the CONSTANT address space
        // its loads are scalar loads (s_load, the scalar cache — what the kernarg-resident objects get), not vector loads of one
        // address by 64 lanes with a vector-memory round trip ahead of every object's arithmetic (measured at 64 objects, 2048²:
        // the table walked with global_load cost the frame 3 x what its instruction count explains — DESIGN.md §4.7).
        typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
        const ConstTable more = (ConstTable)(unsigned long long)sc.more;
        for (uint32_t o = (uint32_t)RTGR_MAX_OBJECTS; o < sc.nobj; o++)
            f(*(const DevObject<R>*)(more + (o - (uint32_t)RTGR_MAX_OBJECTS)), o);
    }
#endif
}
// The same walk with the spheres — objects [0, nsph) of the regrouped list (DevScene) — handed to a function of their own:
// fs(sphere, position) needs no dispatch on the kind, fo(object, position) is the general one.  For consumers that do not care
// about the order (the reach test's conjunction, the minima of the sample-point scan).
template <class R, class FS, class FO>
RTGR_DEV void for_each_by_kind(const DevScene<R>& sc, FS&& fs, FO&& fo) {
    const uint32_t n0 = sc.nobj < (uint32_t)RTGR_MAX_OBJECTS ? sc.nobj : (uint32_t)RTGR_MAX_OBJECTS;
    const uint32_t s0 = sc.nsph < n0 ? sc.nsph : n0;
    for (uint32_t o = 0; o < s0; o++) fs(sc.obj[o], o);
    for (uint32_t o = s0; o < n0; o++) fo(sc.obj[o], o);
#ifndef RTGR_INLINE_OBJECTS_ONLY
    if (__builtin_expect(sc.nobj > (uint32_t)RTGR_MAX_OBJECTS, 0)) {
        typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
        const ConstTable more = (ConstTable)(unsigned long long)sc.more;
        const uint32_t s1 = sc.nsph < sc.nobj ? sc.nsph : sc.nobj;
        for (uint32_t o = (uint32_t)RTGR_MAX_OBJECTS; o < s1; o++) fs(*(const DevObject<R>*)(more + (o - (uint32_t)RTGR_MAX_OBJECTS)), o);
        for (uint32_t o = s1 > (uint32_t)RTGR_MAX_OBJECTS ? s1 : (uint32_t)RTGR_MAX_OBJECTS; o < sc.nobj; o++)
            fo(*(const DevObject<R>*)(more + (o - (uint32_t)RTGR_MAX_OBJECTS)), o);
    }
#endif
}
// The sample-point scan's walk (rtgr_persistent.hpp): the objects whose bit — bit (position >> shift) — is set in `mask`, spheres to fs,
// the other kinds to fo.  A list in the argument block is walked object by object with the bit tested on the way (the hot loop's
// stream); a longer one BY THE SET BITS, through the device table: the scan of a step that can meet three of 100000 objects is not
// a walk over 100000 bits.
template <class R, class FS, class FO>
RTGR_DEV void for_each_masked_by_kind(const DevScene<R>& sc, unsigned long long mask, uint32_t shift, FS&& fs, FO&& fo) {
#ifndef RTGR_INLINE_OBJECTS_ONLY
    if (__builtin_expect(sc.nobj > (uint32_t)RTGR_MAX_OBJECTS, 0)) {
        typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
        const ConstTable table = (ConstTable)(unsigned long long)(sc.more - (uint32_t)RTGR_MAX_OBJECTS);
        unsigned long long m = mask;
        while (m != 0ull) {
            const uint32_t b = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            const uint32_t o0 = b << shift;
            uint32_t o1 = o0 + (1u << shift);
            o1 = o1 < sc.nobj ? o1 : sc.nobj;
            for (uint32_t o = o0; o < o1; o++) {
                if (o < sc.nsph) fs(*(const DevObject<R>*)(table + o), o);
                else fo(*(const DevObject<R>*)(table + o), o);
            }
        }
        return;
    }
#endif
    const uint32_t n0 = sc.nobj < (uint32_t)RTGR_MAX_OBJECTS ? sc.nobj : (uint32_t)RTGR_MAX_OBJECTS;   // (such a list has one bit per object: shift = 0)
    const uint32_t s0 = sc.nsph < n0 ? sc.nsph : n0;
    for (uint32_t o = 0; o < s0; o++) if ((mask >> o) & 1ull) fs(sc.obj[o], o);
    for (uint32_t o = s0; o < n0; o++) if ((mask >> o) & 1ull) fo(sc.obj[o], o);
}
// The reach test's walk (rtgr_persistent.hpp): as for_each_by_kind, but a list with GROUPS (DevScene, rtgr_args.hpp) is walked group
// by group — fg(group, level) -> wave-uniform "some lane cannot rule this group out"; only then its members are handed to fs.  With a
// second level (nsuper > 0) the runs of groups are asked first (level 1), their groups (level 0) only when a run is not ruled out.  The
// whole walk of a grouped list reads the device table (scalar loads through the constant address space, as above) and is cold code for
// every list without groups.
template <class R, class FG, class FS, class FO>
RTGR_DEV void for_each_within_reach(const DevScene<R>& sc, FG&& fg, FS&& fs, FO&& fo) {
#ifndef RTGR_INLINE_OBJECTS_ONLY
    if (__builtin_expect(sc.ngroups != 0u, 0)) {
        typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
        const ConstTable table = (ConstTable)(unsigned long long)(sc.more - (uint32_t)RTGR_MAX_OBJECTS);
        const ConstTable groups = table + sc.nobj;
        const ConstTable supers = groups + sc.ngroups;
        for (uint32_t o = 0; o < sc.nloose; o++) fs(*(const DevObject<R>*)(table + o), o);
        const uint32_t runs = sc.nsuper != 0u ? sc.nsuper : 1u;
        for (uint32_t s = 0; s < runs; s++) {
            uint32_t g0 = 0u, g1 = sc.ngroups;
            if (sc.nsuper != 0u) {
                const DevObject<R>& S = *(const DevObject<R>*)(supers + s);
                if (!fg(S, 1)) continue;
                g0 = S.type;
                g1 = S.type + S.orig;
            }
            for (uint32_t g = g0; g < g1; g++) {
                const DevObject<R>& G = *(const DevObject<R>*)(groups + g);
                if (fg(G, 0)) {
                    const uint32_t o1 = G.type + G.orig;
                    for (uint32_t o = G.type; o < o1; o++) fs(*(const DevObject<R>*)(table + o), o);
                }
            }
        }
        for (uint32_t o = sc.nsph; o < sc.nobj; o++) fo(*(const DevObject<R>*)(table + o), o);
        return;
    }
#endif
    for_each_by_kind<R>(sc, fs, fo);
}
// A SAMPLE of a grouped list (ngroups > 0): the loose spheres, ONE member of every group (of every run of groups, where there are
// runs), the other kinds — f(object, position).
template <class R, class F>
RTGR_DEV void for_each_sample(const DevScene<R>& sc, F&& f) {
    typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
    const ConstTable table = (ConstTable)(unsigned long long)(sc.more - (uint32_t)RTGR_MAX_OBJECTS);
    const ConstTable groups = table + sc.nobj;
    for (uint32_t o = 0; o < sc.nloose; o++) f(*(const DevObject<R>*)(table + o), o);
    if (sc.nsuper != 0u) {   // (with a second level: one member of every RUN of groups — the bound only has to be a bound)
        const ConstTable supers = groups + sc.ngroups;
        for (uint32_t s = 0; s < sc.nsuper; s++) {
            const uint32_t g = ((const DevObject<R>*)(supers + s))->type;
            const uint32_t o = ((const DevObject<R>*)(groups + g))->type;
            f(*(const DevObject<R>*)(table + o), o);
        }
    } else {
        for (uint32_t g = 0; g < sc.ngroups; g++) {
            const uint32_t o = ((const DevObject<R>*)(groups + g))->type;
            f(*(const DevObject<R>*)(table + o), o);
        }
    }
    for (uint32_t o = sc.nsph; o < sc.nobj; o++) f(*(const DevObject<R>*)(table + o), o);
}
// … and one object by (per-lane) POSITION in the regrouped list
template <class R>
RTGR_DEV const DevObject<R>& object_at(const DevScene<R>& sc, uint32_t o) {
    return o < (uint32_t)RTGR_MAX_OBJECTS ? sc.obj[o] : sc.more[o - (uint32_t)RTGR_MAX_OBJECTS];
}

// A SELECTION of the list's objects (the resolve kernel, rtgr_persistent.hpp: select_objects): bit o >> shift of the mask says
// whether object o (position in the device list) takes part.  Only ever used to leave out objects that provably cannot be the minimum.
struct ObjSel {
    unsigned long long mask;
    uint32_t shift;
    // lists beyond 64 objects (a bit is 2, 4, … 2048 neighbours): the selected positions themselves, entry k in lane k of `list`
    // (read with v_readlane, which ignores EXEC), `count` of them; count > 64: too many, walk the mask's blocks
    uint32_t list;
    uint32_t count;
    RTGR_DEV bool has(uint32_t o) const { return ((mask >> (o >> shift)) & 1ull) != 0ull; }
    RTGR_DEV void add(uint32_t o) {   // wave-uniform o
        mask |= 1ull << (o >> shift);
        if (shift != 0u) {
            if (count < 64u) list = ((threadIdx.x & 63u) == count) ? o : list;   // (called with every lane of the wave active: select_objects)
            count++;
        }
    }
};
RTGR_DEV uint32_t objsel_shift(uint32_t nobj) {   // the smallest shift with (nobj − 1) >> shift <= 63
    return nobj > 64u ? 32u - (uint32_t)__builtin_clz((nobj - 1u) >> 6) : 0u;
}

// The objects of a selection, in list order — f(object, position) —, found by the mask's set bits (a list of 1024 objects is not walked
// to find the three that are selected).  Lists beyond the argument block only: the device table holds the whole list.
template <class R, class F>
RTGR_DEV void for_each_selected(const DevScene<R>& sc, ObjSel sel, F&& f) {
    typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
    const ConstTable table = (ConstTable)(unsigned long long)(sc.more - (uint32_t)RTGR_MAX_OBJECTS);
    if (sel.shift != 0u && sel.count <= 64u) {   // by the list of positions
        for (uint32_t k = 0; k < sel.count; k++) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)sel.list, (int)k);
            f(*(const DevObject<R>*)(table + o), o);
        }
        return;
    }
    unsigned long long m = sel.mask;
    while (m != 0ull) {
        const uint32_t b = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        const uint32_t o0 = b << sel.shift;
        uint32_t o1 = o0 + (1u << sel.shift);
        o1 = o1 < sc.nobj ? o1 : sc.nobj;
        for (uint32_t o = o0; o < o1; o++) f(*(const DevObject<R>*)(table + o), o);
    }
}

template <class R, bool SEL = false>
RTGR_DEV R min_distance(const DevScene<R>& sc, const R pos[4], ObjSel sel = ObjSel{}) {   // :433-441
    R dmin = R(__builtin_huge_val());
    auto fold = [&](const DevObject<R>& ob, uint32_t) {
        const R d = obj_distance<R>(ob, pos);
        dmin = (d < dmin || d != d) ? d : dmin;
    };
    if constexpr (SEL) for_each_selected<R>(sc, sel, fold);
    else for_each_object<R>(sc, fold);
    return dmin;
}

}  // namespace rtgr
