// rtgr_persistent.hpp — the production trace kernel: persistent waves, lane refill, batched event resolution.
//
// Why: in the tile-per-wave kernel (trace_kernel, kept as the simple variant) a wave runs until its LONGEST ray ends
// (27…991 step attempts inside one image) and every ray's event episode (root-find, dense output,
// colouring) is executed by the whole wave with one lane active.  Here a wave is a pool of 64 independent ray
// slots:
//   * a lane whose ray has ended parks (state EVENT/DONE); a wavefront ballot counts the parked lanes and when
//     >= `thresh` of them have accumulated they are resolved together (root-find on the dense output, end state,
//     colouring rule, stores) and re-filled from a global ray queue (one wave-aggregated atomic per batch);
//   * the two RHS evaluations of the Hairer initial-step estimate of a fresh ray ride in the k2/k3 evaluation slots
//     of the neighbours' regular Tsit5 step (an "init pseudo-step"), so initialisation costs no extra wave-wide RHS;
//   * the ContinuousCallback's interior sample points use compile-time dense-output weights b_i(j/9).
// The numerical algorithm per ray is unchanged; only the schedule differs.
#pragma once
#include "rtgr_integrator.hpp"

#ifndef RTGR_WAVES_PER_SIMD
#define RTGR_WAVES_PER_SIMD 2  // 2 -> <=256 VGPR+AGPR per lane; 1 -> the whole 512-entry file
#endif

namespace rtgr {

enum LaneState : int { L_FREE = 0, L_INIT = 1, L_RUN = 2, L_EVENT = 3, L_DONE = 4, L_EXIT = 5 };

template <class R>
struct TraceArgs {
    DevScene<R> sc;
    DevSolver<R> opt;
    DevCamera<R> cam;
    const R* state0;  // n x 8 or null (camera)
    uint64_t ni, nj, j0, nrows;
    R* rgb;           // 3 planes of n
    R* state_end;     // optional
    R* lambda_end;
    uint8_t* status;
    uint8_t* hit;
    uint32_t* n_accept;
    uint32_t* n_reject;
    unsigned long long* counters;  // rtgr_counters or null
};

// b_i(θ) for θ = j/(N-1), evaluated at compile time (Tsit5 dense output)
template <class R>
constexpr R dense_w(int i, int j, int nm1) {
    const R th = R(j) / R(nm1);
    return th * (Tsit5C<R>::r[i][0] + th * (Tsit5C<R>::r[i][1] + th * (Tsit5C<R>::r[i][2] + th * Tsit5C<R>::r[i][3])));
}

// position at interior sample point J of NPTS (compile-time weights): x0 + h Σ b_i k_i
template <class R, int J, int NM1>
RTGR_DEV void dense_pos_const(const R y[8], R h, const R k[7][8], R x[4]) {
    constexpr R b0 = dense_w<R>(0, J, NM1), b1 = dense_w<R>(1, J, NM1), b2 = dense_w<R>(2, J, NM1),
                b3 = dense_w<R>(3, J, NM1), b4 = dense_w<R>(4, J, NM1), b5 = dense_w<R>(5, J, NM1),
                b6 = dense_w<R>(6, J, NM1);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        R acc = b0 * k[0][c];
        acc = rfma(b1, k[1][c], acc);
        acc = rfma(b2, k[2][c], acc);
        acc = rfma(b3, k[3][c], acc);
        acc = rfma(b4, k[4][c], acc);
        acc = rfma(b5, k[5][c], acc);
        acc = rfma(b6, k[6][c], acc);
        x[c] = rfma(h, acc, y[c]);
    }
}

template <class R, int J, int NM1>
struct InteriorScan {
    // first J (1..NM1-1) whose sample has the opposite sign of ps; recursion unrolled at compile time
    static RTGR_DEV void run(const DevScene<R>& sc, const R y[8], R h, const R k[7][8], R ps, bool& found, R& top) {
        R xi[4];
        dense_pos_const<R, J, NM1>(y, h, k, xi);
        const R c = min_distance<R>(sc, xi);
        const bool hit = (ps * rsign(c) < R(0)) && !found;
        top = hit ? R(J) / R(NM1) : top;
        found = found || hit;
        if constexpr (J + 1 < NM1) InteriorScan<R, J + 1, NM1>::run(sc, y, h, k, ps, found, top);
    }
};

// polynomial form of the position interpolant: x(θ) = y + θ c1 + θ² c2 + θ³ c3 + θ⁴ c4
template <class R>
RTGR_DEV void dense_pos_coeffs(R h, const R k[7][8], R c[4][4]) {
    using C = Tsit5C<R>;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            R acc = C::r[0][m] * k[0][q];
#pragma unroll
            for (int i = 1; i < 7; i++) acc = rfma(C::r[i][m], k[i][q], acc);
            c[m][q] = h * acc;
        }
}
template <class R>
RTGR_DEV R cond_at(const DevScene<R>& sc, const R y[8], const R c[4][4], R th) {
    R x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = rfma(th, rfma(th, rfma(th, rfma(th, c[3][q], c[2][q]), c[1][q]), c[0][q]), y[q]);
    return min_distance<R>(sc, x);
}

// Bracketed root of cond(x(θ)) on [0, top] with sign(cond(0)) = ps: Ridders' method (quadratic convergence, every
// iterate stays inside the bracket, both ends move) finished by bisection; returns the pre-crossing end of the
// final bracket — the reference's prevfloat(find_zero(...)).
template <class R>
RTGR_DEV R event_root(const DevScene<R>& sc, const R y[8], const R c[4][4], R ps, R top) {
    R lo = R(0), hi = top;
    R fhi = cond_at<R>(sc, y, c, hi) * ps;   // work with g = ps*cond: g(lo) > 0, g(hi) <= 0
    if (fhi == R(0)) return hi;
    R flo = cond_at<R>(sc, y, c, R(0)) * ps;
    if (!(flo > R(0))) return R(0);
    const R eps = sizeof(R) == 8 ? R(2.220446049250313e-16) : R(1.1920929e-7);
    for (int it = 0; it < 100; it++) {
        const R width = hi - lo;
        if (!(width > R(2) * eps * hi)) break;
        const R mid = rfma(R(0.5), width, lo);
        if (!(mid > lo && mid < hi)) break;
        const R fm = cond_at<R>(sc, y, c, mid) * ps;
        // Ridders: x4 = mid + (mid-lo) * sign(flo-fhi) * fm / sqrt(fm² - flo fhi); flo > 0 >= fhi so sign = +1
        const R den = rsqrt_(rfma(fm, fm, -flo * fhi));
        R x4 = den > R(0) ? rfma(mid - lo, fm / den, mid) : mid;
        if (!(x4 > lo && x4 < hi)) x4 = mid;
        const R f4 = (x4 == mid) ? fm : cond_at<R>(sc, y, c, x4) * ps;
        // re-bracket with the tightest pair around the sign change among {lo, mid, x4, hi}
        const R a = rmin(mid, x4), b = rmax(mid, x4);
        const R fa = (mid <= x4) ? fm : f4, fb = (mid <= x4) ? f4 : fm;
        if (fa > R(0)) {
            lo = a; flo = fa;
            if (fb > R(0)) { lo = b; flo = fb; } else { hi = b; fhi = fb; }
        } else {
            hi = a; fhi = fa;
        }
    }
    return lo;
}

// lanes below `lane` set in mask
RTGR_DEV uint32_t mask_rank(unsigned long long mask, uint32_t lane) {
    return (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
}

template <class R, int METRIC, bool SPIN, bool NPTS10>
__global__ __launch_bounds__(64, RTGR_WAVES_PER_SIMD) void trace_persistent_kernel(const TraceArgs<R> A, unsigned long long* queue,
                                                                 int thresh) {
    using C = Tsit5C<R>;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n = A.ni * A.nrows;
    const uint64_t tiles_i = (A.ni + 7) >> 3, tiles_j = (A.nrows + 7) >> 3;
    const uint64_t total = tiles_i * tiles_j * 64;  // work ids in 8x8-tile order (some ids fall outside ragged edges)
    const R M = A.sc.M, aspin = A.sc.a;
    const R reltol = A.opt.reltol, abstol = A.opt.abstol;
    const R t0 = A.opt.lambda0, t1 = A.opt.lambda1, dtmax = A.opt.lambda1 - A.opt.lambda0;
    const R igamma = R(1) / R(0.9L), qmin_inv = R(5), qmax_inv = R(0.1L), qoldinit = R(1e-4L);
    const R beta1 = R(0.14L), beta2 = R(0.08L);
    const int npts = (int)A.opt.interp_points;

    int state = L_FREE;
    bool exhausted = false;
    R y[8], k[7][8];
    R t = t0, dt = R(0), qold = qoldinit, prev_cond = R(0), top = R(1), hstep = R(0);
    uint64_t idx = 0;
    uint32_t nacc = 0, nrej = 0;
    uint8_t status = 0, interior = 0;
    // per-lane totals, reduced once at exit
    unsigned long long c_rays = 0, c_acc = 0, c_rej = 0, c_rhs = 0, c_ev = 0, c_int = 0, c_nf = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        y[i] = R(0);
#pragma unroll
        for (int s = 0; s < 7; s++) k[s][i] = R(0);
    }

    for (;;) {
        // ================= service phase: resolve parked lanes, refill from the ray queue ======================
        const unsigned long long m_wait = __ballot(state == L_EVENT || state == L_DONE);
        const unsigned long long m_free = __ballot(state == L_FREE);
        const unsigned long long m_run = __ballot(state == L_RUN || state == L_INIT);
        const int need = exhausted ? 65 : thresh;
        if (m_free != 0ull || m_run == 0ull || __builtin_popcountll(m_wait) >= need) {
            if (state == L_EVENT || state == L_DONE) {
                R se[8], lam, col[3];
                if (state == L_EVENT) {
                    R cc[4][4];
                    dense_pos_coeffs<R>(hstep, k, cc);
                    const R Theta = event_root<R>(A.sc, y, cc, rsign(prev_cond), top);
                    dense_full<R>(y, hstep, k, Theta, se);
                    lam = rfma(hstep, Theta, t);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) se[i] = y[i];
                    lam = t;
                }
                const uint8_t hit = colour_pixel<R>(A.sc, A.opt, se, col);
                A.rgb[idx] = col[0];
                A.rgb[n + idx] = col[1];
                A.rgb[2 * n + idx] = col[2];
                if (A.state_end) {
#pragma unroll
                    for (int i = 0; i < 8; i++) A.state_end[idx * 8 + i] = se[i];
                }
                if (A.lambda_end) A.lambda_end[idx] = lam;
                if (A.status) A.status[idx] = status;
                if (A.hit) A.hit[idx] = hit;
                if (A.n_accept) A.n_accept[idx] = nacc;
                if (A.n_reject) A.n_reject[idx] = nrej;
                c_rays += 1; c_acc += nacc; c_rej += nrej;
                c_ev += (status == RTGR_RAY_EVENT); c_int += interior; c_nf += (status >= RTGR_RAY_MAXSTEPS);
                state = L_FREE;
            }
            // ---- refill: wave-aggregated pop from the global queue ------------------------------------------
            unsigned long long m_need = __ballot(state == L_FREE);
            while (m_need != 0ull && !exhausted) {  // loops only to skip work ids that fall outside a ragged edge
                const uint32_t cnt = (uint32_t)__builtin_popcountll(m_need);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(queue, (unsigned long long)cnt);
                base = __shfl(base, 0, 64);
                if (state == L_FREE) {
                    const uint64_t w = base + mask_rank(m_need, lane);
                    if (w < total) {
                        const uint64_t tile = w >> 6, l = w & 63;
                        const uint64_t i = (tile % tiles_i) * 8 + (l & 7), jl = (tile / tiles_i) * 8 + (l >> 3);
                        if (i < A.ni && jl < A.nrows) {
                            idx = i + jl * A.ni;
                            if (A.state0) {
#pragma unroll
                                for (int q = 0; q < 8; q++) y[q] = A.state0[idx * 8 + q];
                            } else {
                                make_pixel<R>(A.sc, A.cam, A.ni, A.nj, i, A.j0 + jl, y);
                            }
#pragma unroll
                            for (int q = 0; q < 8; q++) k[0][q] = R(0);
                            t = t0; nacc = 0; nrej = 0; status = 0; interior = 0; top = R(1);
                            state = L_INIT;
                        }
                    }
                }
                if (base + cnt >= total) exhausted = true;
                m_need = __ballot(state == L_FREE);
            }
            if (state == L_FREE) state = L_EXIT;
            if (__ballot(state == L_INIT || state == L_RUN) == 0ull) break;
        }

        // ================= step phase: one Tsit5 attempt (or the init pseudo-step) per runnable lane =============
        const bool init = (state == L_INIT);
        const unsigned long long m_init = __ballot(init);
        if (state == L_RUN || state == L_INIT) {
            if (!init) dt = rmin(dt, t1 - t);
            const R h = init ? R(0) : dt;
            R Y[8], yn[8];
#pragma unroll
            for (int i = 0; i < 8; i++) Y[i] = rfma(h * C::a21, k[0][i], y[i]);
            rhs<R, METRIC, SPIN>(Y, M, aspin, k[1]);            // init lanes: k2 = f(y0) = f0
            R dt0 = R(0), d1 = R(0);
            if (m_init != 0ull) {
                if (init) {  // Hairer initial step, first half
                    R acc0 = R(0), acc1 = R(0);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const R isk = R(1) / rfma(rabs(y[i]), reltol, abstol);
                        const R a0 = y[i] * isk, a1 = k[1][i] * isk;
                        acc0 = rfma(a0, a0, acc0);
                        acc1 = rfma(a1, a1, acc1);
                    }
                    const R d0 = rsqrt_(acc0 * R(0.125));
                    d1 = rsqrt_(acc1 * R(0.125));
                    dt0 = (d0 < R(1e-5) || d1 < R(1e-5)) ? R(1e-6) : (d0 / d1) * R(0.01);
                    dt0 = rmin(dt0, dtmax);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const R yr = rfma(h, rfma(C::a32, k[1][i], C::a31 * k[0][i]), y[i]);
                Y[i] = init ? rfma(dt0, k[1][i], y[i]) : yr;
            }
            rhs<R, METRIC, SPIN>(Y, M, aspin, k[2]);            // init lanes: k3 = f(y0 + dt0 f0) = f1
            R dt_init = R(0);
            if (m_init != 0ull) {
                if (init) {  // second half
                    R acc2 = R(0);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const R isk = R(1) / rfma(rabs(y[i]), reltol, abstol);
                        const R a2 = (k[2][i] - k[1][i]) * isk;
                        acc2 = rfma(a2, a2, acc2);
                    }
                    const R d2 = rsqrt_(acc2 * R(0.125)) / dt0;
                    const R md = rmax(d1, d2);
                    R dt1;
                    if (md <= R(1e-15)) dt1 = rmax(R(1e-6), dt0 * R(1e-3));
                    else dt1 = rpow<R>(R(10), -(R(2) + rlog10<R>(md)) * R(0.2));
                    dt_init = rmin(rmin(R(100) * dt0, dt1), dtmax);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                Y[i] = rfma(h, rfma(C::a43, k[2][i], rfma(C::a42, k[1][i], C::a41 * k[0][i])), y[i]);
            rhs<R, METRIC, SPIN>(Y, M, aspin, k[3]);
#pragma unroll
            for (int i = 0; i < 8; i++)
                Y[i] = rfma(h, rfma(C::a54, k[3][i], rfma(C::a53, k[2][i], rfma(C::a52, k[1][i], C::a51 * k[0][i]))), y[i]);
            rhs<R, METRIC, SPIN>(Y, M, aspin, k[4]);
#pragma unroll
            for (int i = 0; i < 8; i++)
                Y[i] = rfma(h, rfma(C::a65, k[4][i], rfma(C::a64, k[3][i], rfma(C::a63, k[2][i],
                            rfma(C::a62, k[1][i], C::a61 * k[0][i])))), y[i]);
            rhs<R, METRIC, SPIN>(Y, M, aspin, k[5]);
#pragma unroll
            for (int i = 0; i < 8; i++)
                yn[i] = rfma(h, rfma(C::a76, k[5][i], rfma(C::a75, k[4][i], rfma(C::a74, k[3][i], rfma(C::a73, k[2][i],
                             rfma(C::a72, k[1][i], C::a71 * k[0][i]))))), y[i]);
            rhs<R, METRIC, SPIN>(yn, M, aspin, k[6]);

            if (init) {
                // the fresh ray is ready: FSAL slot <- f0, controller state reset            
#pragma unroll
                for (int i = 0; i < 8; i++) k[0][i] = k[1][i];
                dt = dt_init;
                qold = qoldinit;
                prev_cond = min_distance<R>(A.sc, y);
                c_rhs += 2;
                state = L_RUN;
            } else {
                c_rhs += 6;
                R acc = R(0);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const R ut = h * rfma(C::bt7, k[6][i], rfma(C::bt6, k[5][i], rfma(C::bt5, k[4][i],
                                     rfma(C::bt4, k[3][i], rfma(C::bt3, k[2][i], rfma(C::bt2, k[1][i], C::bt1 * k[0][i]))))));
                    const R res = ut / rfma(rmax(rabs(y[i]), rabs(yn[i])), reltol, abstol);
                    acc = rfma(res, res, acc);
                }
                const R EEst = rsqrt_(acc * R(0.125));
                if (EEst != EEst) {
                    status = RTGR_RAY_NAN;
                    state = L_DONE;
                } else {
                    R q, q11 = R(0);
                    if (EEst == R(0)) q = qmax_inv;
                    else {
                        q11 = rpow<R>(EEst, beta1);
                        q = q11 / rpow<R>(qold, beta2);
                        q = rmax(qmax_inv, rmin(qmin_inv, q * igamma));
                    }
                    if (EEst <= R(1)) {
                        nacc++;
                        qold = rmax(EEst, qoldinit);
                        const R dtnew = dt / q;
                        R tnew = t + dt;
                        if (rabs(tnew - t1) < R(10) * R(sizeof(R) == 8 ? 2.220446049250313e-16 : 1.1920929e-7) * rmax(rabs(tnew), rabs(t1)))
                            tnew = t1;
                        // ---- ContinuousCallback ------------------------------------------------
                        const R next_cond = min_distance<R>(A.sc, yn);
                        const R ps = rsign(prev_cond);
                        bool found = false;
                        R tp = R(1);
                        const bool endpoint = (ps != R(0)) && (ps * rsign(next_cond) <= R(0));
                        if constexpr (NPTS10) {
                            InteriorScan<R, 1, 9>::run(A.sc, y, h, k, ps, found, tp);
                        } else {
                            const R dth = npts > 1 ? R(1) / R(npts - 1) : R(1);
                            for (int j = 1; j + 1 < npts; j++) {
                                R xi[4];
                                dense_pos<R>(y, h, k, R(j) * dth, xi);
                                const bool hit = (ps * rsign(min_distance<R>(A.sc, xi)) < R(0)) && !found;
                                tp = hit ? R(j) * dth : tp;
                                found = found || hit;
                            }
                        }
                        found = found && (ps != R(0)) && !endpoint;
                        if (endpoint || found) {
                            top = endpoint ? R(1) : tp;
                            interior = found ? 1 : 0;
                            hstep = h;
                            status = RTGR_RAY_EVENT;
                            state = L_EVENT;  // y, k[0..6], t stay frozen for the batched root-find
                        } else {
                            prev_cond = next_cond;
#pragma unroll
                            for (int i = 0; i < 8; i++) { y[i] = yn[i]; k[0][i] = k[6][i]; }
                            t = tnew;
                            dt = rmin(dtmax, dtnew);
                            if (!(t < t1)) { status = RTGR_RAY_LAMBDA1; state = L_DONE; }
                            else if (nacc + nrej >= A.opt.max_steps) { status = RTGR_RAY_MAXSTEPS; state = L_DONE; }
                            else if (!(t + dt > t)) { status = RTGR_RAY_DTMIN; state = L_DONE; }
                        }
                    } else {
                        nrej++;
                        dt = dt / rmin(qmin_inv, q11 * igamma);
                        if (nacc + nrej >= A.opt.max_steps) { status = RTGR_RAY_MAXSTEPS; state = L_DONE; }
                        else if (!(t + dt > t)) { status = RTGR_RAY_DTMIN; state = L_DONE; }
                    }
                }
            }
        }
    }
    if (A.counters) {
        const unsigned long long s0 = wave_sum(c_rays), s1 = wave_sum(c_acc), s2 = wave_sum(c_rej), s3 = wave_sum(c_rhs),
                                 s4 = wave_sum(c_ev), s5 = wave_sum(c_int), s6 = wave_sum(c_nf);
        if (lane == 0) {
            atomicAdd(&A.counters[0], s0);
            atomicAdd(&A.counters[1], s1);
            atomicAdd(&A.counters[2], s2);
            atomicAdd(&A.counters[3], s3);
            atomicAdd(&A.counters[4], s4);
            atomicAdd(&A.counters[5], s5);
            atomicAdd(&A.counters[6], s6);
        }
    }
}

}  // namespace rtgr
