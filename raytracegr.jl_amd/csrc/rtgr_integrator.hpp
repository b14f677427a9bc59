// rtgr_integrator.hpp — device-side Tsit5 + PI controller + ContinuousCallback for one ray per lane.
//
// Replaces, for the call site src/RayTraceGR.jl:510-511
//     solve(probs, Tsit5(), callback=cb, trajectories=N, reltol=tol, abstol=tol)
// the un-vendored third-party machinery the reference leans on (OrdinaryDiffEq 5.38.3 Tsit5 step + PI controller +
// Hairer initial dt; DiffEqBase 6.35.2 ContinuousCallback with 10 interpolation points; Roots bracketing
// root-find), as restated in SURVEY.md App. A/B.  One wavefront lane owns one ray; state, the 7 stage derivatives
// and the RHS temporaries all live in VGPRs.
#pragma once
#include "rtgr_physics.hpp"

namespace rtgr {

template <class R>
struct Tsit5C {
    static constexpr R a21 = R(0.161L);
    static constexpr R a31 = R(-0.008480655492356989L), a32 = R(0.335480655492357L);
    static constexpr R a41 = R(2.8971530571054935L), a42 = R(-6.359448489975075L), a43 = R(4.3622954328695815L);
    static constexpr R a51 = R(5.325864828439257L), a52 = R(-11.748883564062828L), a53 = R(7.4955393428898365L),
                       a54 = R(-0.09249506636175525L);
    static constexpr R a61 = R(5.86145544294642L), a62 = R(-12.92096931784711L), a63 = R(8.159367898576159L),
                       a64 = R(-0.071584973281401L), a65 = R(-0.028269050394068383L);
    static constexpr R a71 = R(0.09646076681806523L), a72 = R(0.01L), a73 = R(0.4798896504144996L),
                       a74 = R(1.379008574103742L), a75 = R(-3.290069515436081L), a76 = R(2.324710524099774L);
    static constexpr R bt1 = R(-0.00178001105222577714L), bt2 = R(-0.0008164344596567469L),
                       bt3 = R(0.007880878010261995L), bt4 = R(-0.1447110071732629L), bt5 = R(0.5823571654525552L),
                       bt6 = R(-0.45808210592918697L), bt7 = R(0.015151515151515152L);
    // dense output rows r[i][m]: b_i(θ) = Σ_m r[i][m] θ^(m+1)
    static constexpr R r[7][4] = {
        {R(1.0L), R(-2.763706197274826L), R(2.9132554618219126L), R(-1.0530884977290216L)},
        {R(0), R(0.13169999999999998L), R(-0.2234L), R(0.1017L)},
        {R(0), R(3.9302962368947516L), R(-5.941033872131505L), R(2.490627285651252793L)},
        {R(0), R(-12.411077166933676L), R(30.33818863028232L), R(-16.548102889244902L)},
        {R(0), R(37.50931341651104L), R(-88.1789048947664L), R(47.37952196281928L)},
        {R(0), R(-27.896526289197286L), R(65.09189467479366L), R(-34.87065786149661L)},
        {R(0), R(1.5L), R(-4.0L), R(2.5L)}};
};

template <class R> RTGR_DEV R rpow(R x, R y);
template <> RTGR_DEV double rpow<double>(double x, double y) { return pow(x, y); }
template <> RTGR_DEV float rpow<float>(float x, float y) { return powf(x, y); }
template <class R> RTGR_DEV R rlog10(R x);
template <> RTGR_DEV double rlog10<double>(double x) { return log10(x); }
template <> RTGR_DEV float rlog10<float>(float x) { return log10f(x); }

template <class R>
RTGR_DEV R rms8(const R v[8]) {  // ODE_DEFAULT_NORM, SURVEY App. B.1
    R acc = R(0);
#pragma unroll
    for (int i = 0; i < 8; i++) acc = rfma(v[i], v[i], acc);
    return rsqrt_(acc * R(0.125));
}

// b_i(θ), i = 0..6
template <class R>
RTGR_DEV void dense_weights(R th, R b[7]) {
    using C = Tsit5C<R>;
#pragma unroll
    for (int i = 0; i < 7; i++) b[i] = th * rfma(th, rfma(th, rfma(th, C::r[i][3], C::r[i][2]), C::r[i][1]), C::r[i][0]);
}

// position part of the dense output: x(θ) = y0[0..3] + h Σ b_i k_i[0..3]
template <class R>
RTGR_DEV void dense_pos(const R y[8], R h, const R k[7][8], R th, R x[4]) {
    R b[7];
    dense_weights<R>(th, b);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        R acc = b[0] * k[0][c];
#pragma unroll
        for (int i = 1; i < 7; i++) acc = rfma(b[i], k[i][c], acc);
        x[c] = rfma(h, acc, y[c]);
    }
}

template <class R>
RTGR_DEV void dense_full(const R y[8], R h, const R k[7][8], R th, R out[8]) {
    R b[7];
    dense_weights<R>(th, b);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        R acc = b[0] * k[0][c];
#pragma unroll
        for (int i = 1; i < 7; i++) acc = rfma(b[i], k[i][c], acc);
        out[c] = rfma(h, acc, y[c]);
    }
}

struct RayStats {
    uint32_t nacc, nrej, nrhs;
    uint8_t status, interior;
};

// Hairer initial step, SURVEY App. B.3.  f0 = f(y) on entry; uses one more RHS evaluation.
template <class R, int METRIC, bool SPIN>
RTGR_DEV R initial_dt(const R y[8], const R f0[8], R M, R a, R abstol, R reltol, R dtmax) {
    R sk[8], tmp[8];
#pragma unroll
    for (int i = 0; i < 8; i++) sk[i] = R(1) / rfma(rabs(y[i]), reltol, abstol);
#pragma unroll
    for (int i = 0; i < 8; i++) tmp[i] = y[i] * sk[i];
    const R d0 = rms8(tmp);
#pragma unroll
    for (int i = 0; i < 8; i++) tmp[i] = f0[i] * sk[i];
    const R d1 = rms8(tmp);
    R dt0 = (d0 < R(1e-5) || d1 < R(1e-5)) ? R(1e-6) : (d0 / d1) * R(0.01);
    dt0 = rmin(dt0, dtmax);
    R u1[8], f1[8];
#pragma unroll
    for (int i = 0; i < 8; i++) u1[i] = rfma(dt0, f0[i], y[i]);
    rhs<R, METRIC, SPIN>(u1, M, a, f1);
#pragma unroll
    for (int i = 0; i < 8; i++) tmp[i] = (f1[i] - f0[i]) * sk[i];
    const R d2 = rms8(tmp) / dt0;
    const R md = rmax(d1, d2);
    R dt1;
    if (md <= R(1e-15)) dt1 = rmax(R(1e-6), dt0 * R(1e-3));
    else dt1 = rpow<R>(R(10), -(R(2) + rlog10<R>(md)) * R(0.2));
    return rmin(rmin(R(100) * dt0, dt1), dtmax);
}

// One full trajectory.  Returns the end state (`sol[end]`) and λ_end (`sol.t[end]`, src/RayTraceGR.jl:503-504).
template <class R, int METRIC, bool SPIN>
RTGR_DEV RayStats integrate_ray(const DevScene<R>& sc, const DevSolver<R>& opt, const R s0[8], R s_end[8], R& lam_end) {
    using C = Tsit5C<R>;
    const R M = sc.M, a = sc.a;
    const R reltol = opt.reltol, abstol = opt.abstol;
    const R t1 = opt.lambda1;
    const R dtmax = opt.lambda1 - opt.lambda0;
    const R beta1 = R(0.14L), beta2 = R(0.08L), igamma = R(1) / R(0.9L);
    const R qmin_inv = R(5), qmax_inv = R(0.1L), qoldinit = R(1e-4L);
    RayStats st{0, 0, 0, RTGR_RAY_LAMBDA1, 0};

    R y[8], k[7][8];
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = s0[i];
    R t = opt.lambda0;
    rhs<R, METRIC, SPIN>(y, M, a, k[0]);
    R dt = initial_dt<R, METRIC, SPIN>(y, k[0], M, a, abstol, reltol, dtmax);
    st.nrhs = 2;
    R qold = qoldinit;
    R prev_cond = min_distance<R>(sc, y);
    const int npts = (int)opt.interp_points;
    const R dth = npts > 1 ? R(1) / R(npts - 1) : R(1);

    for (;;) {
        if (!(t < t1)) { st.status = RTGR_RAY_LAMBDA1; break; }
        if (st.nacc + st.nrej >= opt.max_steps) { st.status = RTGR_RAY_MAXSTEPS; break; }
        dt = rmin(dt, t1 - t);
        // ---- Tsit5 attempt --------------------------------------------------------------
        R Y[8], yn[8];
        const R h = dt;
#pragma unroll
        for (int i = 0; i < 8; i++) Y[i] = rfma(h * C::a21, k[0][i], y[i]);
        rhs<R, METRIC, SPIN>(Y, M, a, k[1]);
#pragma unroll
        for (int i = 0; i < 8; i++) Y[i] = rfma(h, rfma(C::a32, k[1][i], C::a31 * k[0][i]), y[i]);
        rhs<R, METRIC, SPIN>(Y, M, a, k[2]);
#pragma unroll
        for (int i = 0; i < 8; i++)
            Y[i] = rfma(h, rfma(C::a43, k[2][i], rfma(C::a42, k[1][i], C::a41 * k[0][i])), y[i]);
        rhs<R, METRIC, SPIN>(Y, M, a, k[3]);
#pragma unroll
        for (int i = 0; i < 8; i++)
            Y[i] = rfma(h, rfma(C::a54, k[3][i], rfma(C::a53, k[2][i], rfma(C::a52, k[1][i], C::a51 * k[0][i]))), y[i]);
        rhs<R, METRIC, SPIN>(Y, M, a, k[4]);
#pragma unroll
        for (int i = 0; i < 8; i++)
            Y[i] = rfma(h, rfma(C::a65, k[4][i], rfma(C::a64, k[3][i], rfma(C::a63, k[2][i],
                        rfma(C::a62, k[1][i], C::a61 * k[0][i])))), y[i]);
        rhs<R, METRIC, SPIN>(Y, M, a, k[5]);
#pragma unroll
        for (int i = 0; i < 8; i++)
            yn[i] = rfma(h, rfma(C::a76, k[5][i], rfma(C::a75, k[4][i], rfma(C::a74, k[3][i], rfma(C::a73, k[2][i],
                         rfma(C::a72, k[1][i], C::a71 * k[0][i]))))), y[i]);
        rhs<R, METRIC, SPIN>(yn, M, a, k[6]);
        st.nrhs += 6;
        R acc = R(0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const R ut = h * rfma(C::bt7, k[6][i], rfma(C::bt6, k[5][i], rfma(C::bt5, k[4][i], rfma(C::bt4, k[3][i],
                             rfma(C::bt3, k[2][i], rfma(C::bt2, k[1][i], C::bt1 * k[0][i]))))));
            const R res = ut / rfma(rmax(rabs(y[i]), rabs(yn[i])), reltol, abstol);
            acc = rfma(res, res, acc);
        }
        const R EEst = rsqrt_(acc * R(0.125));
        if (EEst != EEst) { st.status = RTGR_RAY_NAN; break; }
        // ---- PI controller --------------------------------------------------------------
        R q, q11 = R(0);
        if (EEst == R(0)) q = qmax_inv;
        else {
            q11 = rpow<R>(EEst, beta1);
            q = q11 / rpow<R>(qold, beta2);
            q = rmax(qmax_inv, rmin(qmin_inv, q * igamma));
        }
        if (EEst <= R(1)) {
            st.nacc++;
            qold = rmax(EEst, qoldinit);
            const R dtnew = dt / q;
            R tnew = t + dt;
            if (rabs(tnew - t1) < R(10) * R(sizeof(R) == 8 ? 2.220446049250313e-16 : 1.1920929e-7) * rmax(rabs(tnew), rabs(t1)))
                tnew = t1;
            // ---- ContinuousCallback -----------------------------------------------------
            const R next_cond = min_distance<R>(sc, yn);
            const R ps = rsign(prev_cond);
            bool event = false;
            R top = R(1);
            if (ps != R(0)) {
                if (ps * rsign(next_cond) <= R(0)) {
                    event = true;
                } else {
                    for (int i = 2; i <= npts; i++) {
                        const R th = R(i - 1) * dth;
                        R xi[4];
                        dense_pos<R>(y, h, k, th, xi);
                        if (ps * rsign(min_distance<R>(sc, xi)) < R(0)) {
                            event = true;
                            top = th;
                            st.interior = (i != npts);
                            break;
                        }
                    }
                }
            }
            if (event) {
                // bracketed root of cond(dense(θ)) on [0, top]; Θ ends on the pre-crossing side (prevfloat)
                R lo = R(0), hi = top, xi[4];
                dense_pos<R>(y, h, k, hi, xi);
                R Theta;
                if (min_distance<R>(sc, xi) == R(0)) Theta = hi;
                else {
                    for (int it = 0; it < 200; it++) {
                        const R mid = rfma(R(0.5), hi - lo, lo);
                        if (!(mid > lo && mid < hi)) break;
                        dense_pos<R>(y, h, k, mid, xi);
                        const R sg = rsign(min_distance<R>(sc, xi));
                        if (sg * ps > R(0)) lo = mid; else hi = mid;
                    }
                    Theta = lo;
                }
                dense_full<R>(y, h, k, Theta, s_end);
                lam_end = rfma(h, Theta, t);
                st.status = RTGR_RAY_EVENT;
                return st;
            }
            prev_cond = next_cond;
#pragma unroll
            for (int i = 0; i < 8; i++) { y[i] = yn[i]; k[0][i] = k[6][i]; }
            t = tnew;
            dt = rmin(dtmax, dtnew);
            if (t < t1 && !(t + dt > t)) { st.status = RTGR_RAY_DTMIN; break; }
        } else {
            st.nrej++;
            dt = dt / rmin(qmin_inv, q11 * igamma);
            if (!(t + dt > t)) { st.status = RTGR_RAY_DTMIN; break; }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) s_end[i] = y[i];
    lam_end = t;
    return st;
}

// ---- colouring rule of trace_rays (src/RayTraceGR.jl:513-533) + objcolor (:402-404, :420-428) --------------------
template <class R> RTGR_DEV R racos(R x);
template <> RTGR_DEV double racos<double>(double x) { return acos(x); }
template <> RTGR_DEV float racos<float>(float x) { return acosf(x); }
template <class R> RTGR_DEV R ratan2(R y, R x);
template <> RTGR_DEV double ratan2<double>(double y, double x) { return atan2(y, x); }
template <> RTGR_DEV float ratan2<float>(float y, float x) { return atan2f(y, x); }
template <class R> RTGR_DEV R rfloor(R x);
template <> RTGR_DEV double rfloor<double>(double x) { return floor(x); }
template <> RTGR_DEV float rfloor<float>(float x) { return floorf(x); }

template <class R>
RTGR_DEV R mod1(R x) {  // Julia mod(x, 1)
    R r = x - rfloor<R>(x);
    return r >= R(1) ? R(0) : r;
}

template <class R, bool SEL = false>
RTGR_DEV uint32_t colour_pixel(const DevScene<R>& sc, const DevSolver<R>& opt, const R x[4], R col[3], ObjSel sel = ObjSel{}) {
    uint32_t omin = 0, pmin = 0;   // omin: 1-based index in the CALLER's list (what :518-530 calls omin); pmin: position in the device list
    R dmin = opt.hit_threshold;                                                       // :519
    // (the device list is regrouped — spheres first, DevScene —: "the first object with the smallest distance wins" (:520-526) is
    //  the smallest distance and, among equal ones, the smallest ORIGINAL index)
    auto nearest = [&](const DevObject<R>& o_, uint32_t o) {                          // :520-526
        const R d = obj_distance<R>(o_, x);
        if (d < dmin || (d == dmin && omin != 0u && o_.orig + 1u < omin)) { omin = o_.orig + 1u; pmin = o; dmin = d; }
    };
    if constexpr (SEL) for_each_selected<R>(sc, sel, nearest);   // (the rest: provably farther than the nearest object, select_objects)
    else for_each_object<R>(sc, nearest);
    if (omin == 0) {                                                                  // :527-528
        col[0] = opt.miss_rgb[0]; col[1] = opt.miss_rgb[1]; col[2] = opt.miss_rgb[2];
        return 0;
    }
    // (a selection exists for lists beyond the argument block only: their table holds the whole list)
    const DevObject<R>& ob = SEL ? (sc.more - (uint32_t)RTGR_MAX_OBJECTS)[pmin] : object_at<R>(sc, pmin);
    const R pi = R(3.14159265358979323846264338327950288L);
    if (ob.kind == RTGR_PLANE) {                                                      // :402-404
        col[0] = R(0); col[1] = R(0.5); col[2] = R(0);
    } else if (ob.kind == RTGR_SPHERE) {                                              // :420-428
        const R dx = x[1] - ob.p[1], dy = x[2] - ob.p[2], dz = x[3] - ob.p[3];
        const R r = rsqrt_(dx * dx + dy * dy + dz * dz);
        const R th = racos<R>(dz / r);
        const R ph = ratan2<R>(dy, dx);
        col[0] = mod1<R>(R(12) * th / pi);
        col[1] = mod1<R>(R(12) * ph / pi);
        col[2] = R(1);
#ifdef RTGR_USER_OBJECTS
    } else if (ob.kind == RTGR_USER_OBJECT) {                                         // objcolor(obj::MyThing, pos)  :387-389
        rtgr_user_objcolor<R>(ob.type, x, ob.p, col);
#endif
    } else {  // RTGR_DISK — no reference counterpart
        const R rc = rsqrt_(x[1] * x[1] + x[2] * x[2]);
        const R ph = ratan2<R>(x[2], x[1]);
        col[0] = R(1);
        col[1] = mod1<R>(rc);
        col[2] = mod1<R>(R(12) * ph / pi);
    }
    const R scale = R(omin) / R(sc.nobj);                                             // :530
    col[0] *= scale; col[1] *= scale; col[2] *= scale;
    return omin;
}

// parity hook rtgr_eval_objects_f64 / _f32: distance(obj, x) of every object (:377-419), min_distance (:433-441) and the colour rule
// (:513-533) at one point per thread — a body function: a unit with user objects wraps it in a kernel of its own
template <class R>
RTGR_DEV void eval_objects_body(const DevScene<R>& sc, const DevSolver<R>& opt, const R* x, uint64_t n, R* d, R* dmin, uint8_t* hit, R* rgb) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const R xp[4] = {x[4 * p], x[4 * p + 1], x[4 * p + 2], x[4 * p + 3]};
    if (d) for_each_object<R>(sc, [&](const DevObject<R>& ob, uint32_t) { d[p * sc.nobj + ob.orig] = obj_distance<R>(ob, xp); });   // (scene order)
    if (dmin) dmin[p] = min_distance<R>(sc, xp);
    R col[3];
    const uint32_t h = colour_pixel<R>(sc, opt, xp, col);
    if (hit) hit[p] = (uint8_t)h;   // (the host side refuses `hit` for lists beyond 255 objects)
    if (rgb) for (int c = 0; c < 3; c++) rgb[3 * p + c] = col[c];
}

// metric(x) with plain scalars (no duals): what make_canvas calls (:469)
template <class R>
RTGR_DEV void metric_plain(const DevScene<R>& sc, const R x[4], R g[4][4]) {
#pragma clang fp contract(off)   // see make_pixel
#ifdef RTGR_USER_METRIC
    if (sc.metric == (uint32_t)RTGR_USER) {
        rtgr_user_metric<R>(x, (double)sc.M, (double)sc.a, g);
        return;
    }
#endif
    // built-ins are η + f k k
    R f = R(0), kk[4] = {R(1), R(0), R(0), R(0)};
    if (sc.metric != RTGR_MINKOWSKI) {
        KSField<R> F;
        if (sc.metric == RTGR_KS_REF) ks_field<R, RTGR_KS_REF, true>(x[1], x[2], x[3], sc.M, sc.a, F);
        else ks_field<R, RTGR_KS_TRUE, true>(x[1], x[2], x[3], sc.M, sc.a, F);
        f = F.f; kk[1] = F.k[0]; kk[2] = F.k[1]; kk[3] = F.k[2];
    }
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) g[p][q] = (p == q ? (p == 0 ? R(-1) : R(1)) : R(0)) + f * kk[p] * kk[q];
}

// ---- make_canvas pixel (src/RayTraceGR.jl:464-476): state (x, u) of pixel (i, j), 0-based -------------------------
template <class R>
RTGR_DEV void make_pixel(const DevScene<R>& sc, const DevCamera<R>& cam, uint64_t ni, uint64_t nj, uint64_t i0,
                         uint64_t j0, R s[8]) {
    // No implicit contraction in the camera (here, metric_plain, ks_field, inv4sym): this function is inlined into
    // prepare_kernel — next to the RHS of the same point — AND into canvas_kernel, and -ffp-contract=fast decides fusions by
    // use counts after inlining and CSE: a product shared with the neighbouring code (a², ρ²) fused in one kernel and not in
    // the other, and the camera ray generated inside the pipeline differed from rtgr_make_canvas' in the last bit (round 3:
    // found by test_host_pipeline_with_many_chunks_and_every_output when the spin RHS changed).  Explicit rfma() stay FMAs.
#pragma clang fp contract(off)
    const R dx = (R(i0 + 1) - R(0.5)) / R(ni) - R(0.5);                               // :465
    const R dy = (R(j0 + 1) - R(0.5)) / R(nj) - R(0.5);                               // :466
    R x[4], n[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        x[c] = cam.pos[c] + dx * cam.widthx[c] + dy * cam.widthy[c];                  // :467
        n[c] = cam.normal[c] + dx * cam.widthx[c] + dy * cam.widthy[c];               // :468
    }
    R g[4][4];
    metric_plain<R>(sc, x, g);                                                        // :469
    R gu[4][4];
    inv4sym<R>(g, gu);                                                                // :470
    R t[4];
#pragma unroll
    for (int p = 0; p < 4; p++) t[p] = gu[p][0];                                      // gu * e_t   :471
    R t2 = R(0), n2 = R(0);
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            t2 += t[p] * g[p][q] * t[q];                                              // :472
            n2 += n[p] * g[p][q] * n[q];                                              // :473
        }
    const R st = rsqrt_(-t2), sn = rsqrt_(n2), s2 = rsqrt_(R(2));
#pragma unroll
    for (int p = 0; p < 4; p++) {
        s[p] = x[p];
        s[4 + p] = (t[p] / st + n[p] / sn) / s2;                                      // :474
    }
}

// ---- redshift (rtgr_ray_outputs.redshift; SURVEY §8 f4: "Doppler/redshift via the unused Sphere.vel", :411, :416) --------
// g = (k·u_obs) / (k·u_emit): the ratio observed / emitted frequency of the light that reaches a pixel.
//   k      = tangent of the traced ray (an affinely parametrised null geodesic, so k is parallel-transported and the
//            ratio does not depend on its normalisation or on the direction the ray was traced in)
//   u_obs  = the static observer make_canvas builds every ray from, future-directed: −g^{-1} e_t / sqrt(−g(t,t)) at the
//            pixel (:471-472; the reference uses the past-directed sign because it traces rays backwards in time)
//   u_emit = Sphere: its `vel` (coordinate 4-velocity as stored in the reference's struct) normalised with the metric at
//            the hit point; Plane / Disk: the static observer t̂ there
//   ·      = the metric at the respective end of the ray — ANY metric: built-in or run-time compiled, Float64 or Float32
// NaN where nothing is hit, or where u_emit is not timelike (a static emitter inside the ergoregion, vel = 0, …).
template <class R>
RTGR_DEV void static_observer(const R g[4][4], R t[4], bool& ok) {
    R gu[4][4];
    inv4sym<R>(g, gu);
    R t2 = R(0);
    for (int p = 0; p < 4; p++) t[p] = gu[p][0];
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < 4; q++) t2 += t[p] * g[p][q] * t[q];
    ok = t2 < R(0);
    const R s = R(-1) / rsqrt_(-t2);   // g^{-1} e_t points to the past (make_canvas builds past-directed rays from it); −: future
    for (int p = 0; p < 4; p++) t[p] *= s;
}
template <class R>
RTGR_DEV R inner(const R g[4][4], const R a[4], const R b[4]) {
    R acc = R(0);
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < 4; q++) acc += a[p] * g[p][q] * b[q];
    return acc;
}
// one thread per ray; a body function so that run-time compiled metric units wrap it in kernels of their own (metric_plain
// dispatches to the unit's rtgr_user_metric there)
template <class R>
RTGR_DEV void redshift_body(const DevScene<R>& sc, const DevCamera<R>& cam, const R* state0, uint64_t ni, uint64_t nj, uint64_t j0,
                            uint64_t jstride, uint64_t n, uint64_t out_offset, const R* state_end, const uint8_t* hit, const uint32_t* hit32, R* red) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const uint64_t idx = out_offset + w;
    const R nan = R(__builtin_nan(""));
    const uint32_t h = hit32 ? hit32[idx] : (uint32_t)hit[idx];
    if (h == 0 || h > sc.nobj) { red[idx] = nan; return; }
    R s0[8], se[8];
    if (state0) for (int c = 0; c < 8; c++) s0[c] = state0[w * 8 + c];
    else make_pixel<R>(sc, cam, ni, nj, w % ni, j0 + (w / ni) * jstride, s0);
    for (int c = 0; c < 8; c++) se[c] = state_end[idx * 8 + c];
    R g0[4][4], ge[4][4], tobs[4], uem[4];
    bool ok0, oke;
    metric_plain<R>(sc, s0, g0);
    static_observer<R>(g0, tobs, ok0);
    metric_plain<R>(sc, se, ge);
    uint32_t pos = 0;                      // the hit map holds indices of the CALLER's list: find the object in the regrouped one
    for_each_object<R>(sc, [&](const DevObject<R>& o_, uint32_t o) { if (o_.orig + 1u == h) pos = o; });
    const DevObject<R>& ob = object_at<R>(sc, pos);
    if (ob.kind == RTGR_SPHERE) {
        const R v[4] = {ob.p[4], ob.p[5], ob.p[6], ob.p[7]};
        const R v2 = inner<R>(ge, v, v);
        oke = v2 < R(0);
        const R s = R(1) / rsqrt_(-v2);
        for (int p = 0; p < 4; p++) uem[p] = v[p] * s;
    } else {
        static_observer<R>(ge, uem, oke);
    }
    const R num = inner<R>(g0, s0 + 4, tobs), den = inner<R>(ge, se + 4, uem);
    red[idx] = (ok0 && oke) ? num / den : nan;
}

}  // namespace rtgr
