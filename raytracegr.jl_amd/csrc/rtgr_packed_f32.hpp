// rtgr_packed_f32.hpp — the Float32 integrate pass with TWO RAYS PER LANE (BASELINE config 4, "register-pressure / LDS-tiled
// Christoffel variant, tolerance relaxed"; `T = Float32` is first-class in the reference: trace_rays is generic in T,
// src/RayTraceGR.jl:483-485, and its own tests run T = Float32, test/runtests.jl:37-60).
//
// Why: a scalar v_fma_f32 issues at the rate of a v_fma_f64 on gfx950 — the 157.3 TFLOP/s fp32 vector peak belongs to
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which do two f32 operations per lane per issue slot.  The scalar Float32 kernel
// (integrate_body<float>) holds one ray per lane; LLVM's SLP vectoriser pairs what it can find inside one ray (components of
// the stage sums: 24 % of the arithmetic instructions), the RHS — scalar by nature — stays scalar.  Here a lane holds rays A
// and B in the two halves of float2 registers and EVERY arithmetic instruction of the step — the six RHS evaluations, stage
// sums, error sums, dense-output polynomial, the nine sample points and their sphere distances — is a packed instruction
// serving both; what has no packed form (reciprocal / rsq / log / exp seeds, min / max, compares, selects, the per-ray
// bookkeeping and record writes) runs once per half.  Same algorithm, same operations per ray in the same order as the
// scalar kernel (a packed FMA is an FMA per half): tests compare the two frame for frame.
//
// Shape: the single FULL pass of the Float32 pipeline (rays last ~21 step attempts at tol = eps^(3/4) = 6.4e-6, so the
// FAR / NEAR split does not pay, rtgr_pipeline.hpp), interp_points = 10, closed-form RHS of the built-in metrics.  Everything
// else (user metrics, generic RHS, other interp_points) keeps the scalar kernel; option pack = 0 forces it for A/B.
#pragma once
#include "rtgr_persistent.hpp"

namespace rtgr {

// ---- float2 versions of the arithmetic helpers the physics templates are written with -------------------------------
using V2 = float2_t;
template <> RTGR_DEV V2 rfma<V2>(V2 a, V2 b, V2 c) { return __builtin_elementwise_fma(a, b, c); }
template <> RTGR_DEV V2 rsqrt_<V2>(V2 x) { return V2{__builtin_sqrtf(x.x), __builtin_sqrtf(x.y)}; }
template <> RTGR_DEV V2 rabs<V2>(V2 x) { return __builtin_elementwise_abs(x); }
template <> RTGR_DEV V2 rmax<V2>(V2 a, V2 b) { return V2{rmax<float>(a.x, b.x), rmax<float>(a.y, b.y)}; }
template <> RTGR_DEV V2 rmin<V2>(V2 a, V2 b) { return V2{rmin<float>(a.x, b.x), rmin<float>(a.y, b.y)}; }
template <> RTGR_DEV V2 rmaxabs<V2>(V2 a, V2 b) { return V2{rmaxabs<float>(a.x, b.x), rmaxabs<float>(a.y, b.y)}; }
template <> RTGR_DEV V2 frcp<V2>(V2 x) {            // seeds per half, the Newton step packed
    const V2 r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    const V2 e = __builtin_elementwise_fma(-x, r, V2(1.0f));
    return __builtin_elementwise_fma(r, e, r);
}
template <> RTGR_DEV V2 frsq<V2>(V2 x) {
    const V2 r = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const V2 e = __builtin_elementwise_fma(-x * r, r, V2(1.0f));
    return __builtin_elementwise_fma(V2(0.5f) * r, e, r);
}
RTGR_DEV V2 rcp_seed2(V2 x) { return V2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
RTGR_DEV MetricK<V2> splat_consts(const MetricK<float>& c) {
    MetricK<V2> k;
    k.M = V2(c.M); k.a = V2(c.a); k.M2 = V2(c.M2); k.a2 = V2(c.a2); k.a2x2 = V2(c.a2x2);
    return k;
}

// distances of one object at P sample positions of BOTH rays folded into dmin[] (object-major, as fold_distances)
template <int P>
RTGR_DEV void fold_distances2(const DevObject<float>& o, const V2 (&pos)[P][4], V2 (&dmin)[P]) {
    if (o.kind == RTGR_PLANE) {                                                        // src/RayTraceGR.jl:399-401
        const V2 tm = V2(o.p[0]);
#pragma unroll
        for (int p = 0; p < P; p++) dmin[p] = rmin<V2>(dmin[p], pos[p][0] - tm);
    } else if (o.kind == RTGR_SPHERE) {                                                // :415-419
        const V2 cx = V2(o.p[1]), cy = V2(o.p[2]), cz = V2(o.p[3]);
        const float Rr = o.p[8];
        const V2 nR2 = V2(-Rr * Rr);
        if (Rr < 0.0f) {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const V2 dx = pos[p][1] - cx, dy = pos[p][2] - cy, dz = pos[p][3] - cz;
                dmin[p] = rmin<V2>(dmin[p], -rfma<V2>(dx, dx, rfma<V2>(dy, dy, rfma<V2>(dz, dz, nR2))));
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const V2 dx = pos[p][1] - cx, dy = pos[p][2] - cy, dz = pos[p][3] - cz;
                dmin[p] = rmin<V2>(dmin[p], rfma<V2>(dx, dx, rfma<V2>(dy, dy, rfma<V2>(dz, dz, nR2))));
            }
        }
    } else {   // RTGR_DISK: per half through the scalar sign-exact surrogate (fold_distances; the asm barrier keeps it inside this branch)
#pragma unroll
        for (int p = 0; p < P; p++) {
            V2 px = pos[p][1], py = pos[p][2];
            asm volatile("" : "+v"(px), "+v"(py));
            dmin[p] = rmin<V2>(dmin[p], V2{disk_sign_distance<float>(o, px.x, py.x, pos[p][3].x),
                                           disk_sign_distance<float>(o, px.y, py.y, pos[p][3].y)});
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the FULL pass, two rays per lane.  Structure and comments follow integrate_body (rtgr_persistent.hpp), MODE_FULL, NPTS10.
// ---------------------------------------------------------------------------------------------------------------------
// FARP = true (round 4, experiment behind option `packfar`): the scan-free FAR instantiation.  The nine sample positions and
// distances of both rays and the position polynomial are what push this kernel from 159 to 205 registers (two waves per SIMD,
// where a wave issues at most 83 % of the slots); without them it fits three.  The scan is replaced, as in the Float64 FAR pass,
// by the rigorous reach bound; a half whose accepted step is not provably clear of every object is handed — in its pre-step
// state — to the scalar NEAR pass (integrate_kernel<float, …, MODE_NEAR>), which redoes that step with the full scan.
template <int METRIC, bool SPIN, bool FARP = false>
RTGR_DEV void integrate2_body(const IntegrateArgs<float>& A) {
    using N = Tsit5N<float>;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t total = A.n;
    unsigned long long* const queue = A.ctrl;
    const MetricK<V2> MK = splat_consts(metric_consts<float>(A.sc.M, A.sc.a));
    const float reltol = A.opt.reltol, abstol = A.opt.abstol;
    const float t0 = A.opt.lambda0, t1 = A.opt.lambda1, dtmax = uniform_(A.opt.lambda1 - A.opt.lambda0);
    const float igamma = 1.0f / 0.9f, qmin_inv = 5.0f, qmax_inv = 0.1f;
    const float lq_init = -13.287712379549449f;  // log2(qoldinit = 1e-4)
    const float beta1 = 0.14f, beta2 = 0.08f;
    const float eps = 1.1920929e-7f;

    int state[2] = {L_FREE, L_FREE};
    bool exhausted = false;
    V2 x[4], u[4], k0[4];                 // loop-carried ray states: half .x = ray A, half .y = ray B
    V2 t = V2(t0), dt = V2(0.0f), ps = V2(0.0f);
    float lq[2] = {lq_init, lq_init};
    uint64_t idx[2] = {0, 0};
    uint32_t nacc[2] = {0, 0}, nrej[2] = {0, 0};
    uint32_t c_rays = 0, c_acc = 0, c_rej = 0, c_ev = 0, c_int = 0, c_nf = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { x[q] = V2(1.0f); u[q] = V2(0.0f); k0[q] = V2(0.0f); }

    uint64_t q_next = 0, q_end = 0;
    bool first_pop[2] = {true, true};
    const unsigned long long qchunk = A.queue_chunk;
    const unsigned long long first_span = 128ull * gridDim.x;   // a wave starts 128 rays: positions [128 b, 128 b + 128) go to workgroup b
    for (;;) {
        // [budget: refill]
        // ================= refill: free slots (half 0, then half 1) take ray ids from the wave's slice of the queue ========
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            unsigned long long m_need = __ballot(state[hh] == L_FREE);
            while (m_need != 0ull) {
                if (q_next == q_end) {
                    if (exhausted) break;
                    // first pops by WAVE INDEX (see "wave ages", rtgr_persistent.hpp): queue positions [128 b, 128 b + 128) start in
                    // workgroup b — 64 per half —, the atomic head serves positions from 128 * gridDim.x on
                    const unsigned long long amount = first_pop[hh] ? 64ull : qchunk;
                    unsigned long long base = 0;
                    if (first_pop[hh]) base = 128ull * blockIdx.x + 64ull * hh;
                    else if (lane == 0) base = atomicAdd(queue, amount) + first_span;
                    first_pop[hh] = false;
                    base = uniform64(base);
                    q_next = base < total ? base : total;
                    q_end = (base + amount) < total ? (base + amount) : total;
                    if (base + amount >= total) exhausted = true;
                    if (q_next == q_end) break;
                }
                const uint64_t avail = q_end - q_next;
                const uint32_t rank = mask_rank(m_need, lane);
                if (state[hh] == L_FREE && rank < avail) {
                    const uint64_t w = q_next + rank;
                    idx[hh] = A.order ? (uint64_t)A.order[w] : w;
                    state[hh] = L_TAKEN;
                }
                const uint32_t cnt = (uint32_t)__builtin_popcountll(m_need);
                q_next += (cnt < avail) ? cnt : avail;
                m_need = __ballot(state[hh] == L_FREE);
            }
            if (state[hh] == L_TAKEN) {
                const float* hd = A.hand + idx[hh] * HAND_W;
#pragma unroll
                for (int q = 0; q < 4; q++) { x[q][hh] = hd[q]; u[q][hh] = hd[4 + q]; k0[q][hh] = hd[8 + q]; }
                t[hh] = hd[12]; dt[hh] = hd[13]; ps[hh] = hd[14]; lq[hh] = hd[15];
                nacc[hh] = 0u; nrej[hh] = 0u;
                state[hh] = L_RUN;
            }
        }
        // The refill loops only give up when the queue is exhausted, so a wave without a running slot is done.
        if (__ballot(state[0] == L_RUN || state[1] == L_RUN) == 0ull) break;

        // [budget: stage sums]
        // ================= one Tsit5 attempt per runnable slot, both halves in packed arithmetic =============================
        const bool run[2] = {state[0] == L_RUN, state[1] == L_RUN};
        V2 xn[4], un[4], k[7][4];
#pragma unroll
        for (int q = 0; q < 4; q++) k[0][q] = k0[q];
        dt = rmin<V2>(dt, V2(t1) - t);                       // (a slot without a ray carries h = 0 below)
        const V2 h = V2{run[0] ? dt.x : 0.0f, run[1] ? dt.y : 0.0f};
        const V2 h2 = h * h;
        {
            V2 X[3], U[4];
            {   // ---- stage 2
                const V2 ha = h * N::a[1][0], hc = h * N::c[1];
#pragma unroll
                for (int q = 0; q < 4; q++) U[q] = rfma<V2>(ha, k[0][q], u[q]);
#pragma unroll
                for (int q = 0; q < 3; q++) X[q] = rfma<V2>(hc, u[1 + q], x[1 + q]);
            }
            // [budget: rhs]
            accel<V2, METRIC, SPIN, true>(X, U, MK, k[1]);
            // [budget: stage sums]
            {   // ---- stage 3
                const V2 w1 = h * N::a[2][1], w0 = h * N::a[2][0], hc = h * N::c[2], h2a = h2 * N::A2[2][0];
#pragma unroll
                for (int q = 0; q < 4; q++) U[q] = rfma<V2>(w1, k[1][q], rfma<V2>(w0, k[0][q], u[q]));
#pragma unroll
                for (int q = 0; q < 3; q++) X[q] = rfma<V2>(h2a, k[0][1 + q], rfma<V2>(hc, u[1 + q], x[1 + q]));
            }
            // [budget: rhs]
            accel<V2, METRIC, SPIN, true>(X, U, MK, k[2]);
            // [budget: stage sums]
            // ---- stages 4, 5, 6
#pragma unroll
            for (int q = 0; q < 4; q++)
                U[q] = rfma<V2>(h, rfma<V2>(V2(N::a[3][2]), k[2][q], rfma<V2>(V2(N::a[3][1]), k[1][q], N::a[3][0] * k[0][q])), u[q]);
#pragma unroll
            for (int q = 0; q < 3; q++)
                X[q] = rfma<V2>(h2, rfma<V2>(V2(N::A2[3][1]), k[1][1 + q], N::A2[3][0] * k[0][1 + q]),
                                rfma<V2>(h * N::c[3], u[1 + q], x[1 + q]));
            // [budget: rhs]
            accel<V2, METRIC, SPIN, true>(X, U, MK, k[3]);
            // [budget: stage sums]
#pragma unroll
            for (int q = 0; q < 4; q++)
                U[q] = rfma<V2>(h, rfma<V2>(V2(N::a[4][3]), k[3][q], rfma<V2>(V2(N::a[4][2]), k[2][q], rfma<V2>(V2(N::a[4][1]), k[1][q],
                                N::a[4][0] * k[0][q]))), u[q]);
#pragma unroll
            for (int q = 0; q < 3; q++)
                X[q] = rfma<V2>(h2, rfma<V2>(V2(N::A2[4][2]), k[2][1 + q], rfma<V2>(V2(N::A2[4][1]), k[1][1 + q], N::A2[4][0] * k[0][1 + q])),
                                rfma<V2>(h * N::c[4], u[1 + q], x[1 + q]));
            // [budget: rhs]
            accel<V2, METRIC, SPIN, true>(X, U, MK, k[4]);
            // [budget: stage sums]
#pragma unroll
            for (int q = 0; q < 4; q++)
                U[q] = rfma<V2>(h, rfma<V2>(V2(N::a[5][4]), k[4][q], rfma<V2>(V2(N::a[5][3]), k[3][q], rfma<V2>(V2(N::a[5][2]), k[2][q],
                                rfma<V2>(V2(N::a[5][1]), k[1][q], N::a[5][0] * k[0][q])))), u[q]);
#pragma unroll
            for (int q = 0; q < 3; q++)
                X[q] = rfma<V2>(h2, rfma<V2>(V2(N::A2[5][3]), k[3][1 + q], rfma<V2>(V2(N::A2[5][2]), k[2][1 + q], rfma<V2>(V2(N::A2[5][1]), k[1][1 + q],
                                N::A2[5][0] * k[0][1 + q]))), rfma<V2>(h * N::c[5], u[1 + q], x[1 + q]));
            // [budget: rhs]
            accel<V2, METRIC, SPIN, true>(X, U, MK, k[5]);
            // [budget: stage sums]
            // ---- stage 7 = the step result (FSAL)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                un[q] = rfma<V2>(h, rfma<V2>(V2(N::a[6][5]), k[5][q], rfma<V2>(V2(N::a[6][4]), k[4][q], rfma<V2>(V2(N::a[6][3]), k[3][q],
                                 rfma<V2>(V2(N::a[6][2]), k[2][q], rfma<V2>(V2(N::a[6][1]), k[1][q], N::a[6][0] * k[0][q]))))), u[q]);
                xn[q] = rfma<V2>(h2, rfma<V2>(V2(N::A2[6][4]), k[4][q], rfma<V2>(V2(N::A2[6][3]), k[3][q], rfma<V2>(V2(N::A2[6][2]), k[2][q],
                                 rfma<V2>(V2(N::A2[6][1]), k[1][q], N::A2[6][0] * k[0][q])))), rfma<V2>(h * N::c[6], u[q], x[q]));
            }
            // [budget: rhs]
            accel<V2, METRIC, SPIN, true>(xn + 1, un, MK, k[6]);
        }
        // [budget: error norm]
        // ---- embedded error: both rays' sums packed, the two reciprocal seeds per component per half ----
        V2 acc2 = V2(0.0f);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const V2 eu = rfma<V2>(V2(N::bt[6]), k[6][q], rfma<V2>(V2(N::bt[5]), k[5][q], rfma<V2>(V2(N::bt[4]), k[4][q],
                              rfma<V2>(V2(N::bt[3]), k[3][q], rfma<V2>(V2(N::bt[2]), k[2][q], rfma<V2>(V2(N::bt[1]), k[1][q], N::bt[0] * k[0][q]))))));
            const V2 ex = rfma<V2>(h, rfma<V2>(V2(N::BT2[5]), k[5][q], rfma<V2>(V2(N::BT2[4]), k[4][q], rfma<V2>(V2(N::BT2[3]), k[3][q],
                              rfma<V2>(V2(N::BT2[2]), k[2][q], rfma<V2>(V2(N::BT2[1]), k[1][q], N::BT2[0] * k[0][q]))))), N::sbt * u[q]);
            const V2 ru = eu * rcp_seed2(rfma<V2>(rmaxabs<V2>(u[q], un[q]), V2(reltol), V2(abstol)));
            const V2 rx = ex * rcp_seed2(rfma<V2>(rmaxabs<V2>(x[q], xn[q]), V2(reltol), V2(abstol)));
            acc2 = rfma<V2>(ru, ru, rfma<V2>(rx, rx, acc2));
        }
        const V2 EE2 = (acc2 * h) * (V2(0.125f) * h);   // (h twice, not h²: see integrate_body)
        // [budget: scan]
        // ---- ContinuousCallback scan, packed for both rays; used by the halves that accept their step ----
        // x(θ) = x + θ c1 + θ² c2 + θ³ c3 + θ⁴ c4 ;  c1 = h u,  c_m = h² Σ_l R2[l][m] k_l
        V2 cc[4][4];
        V2 nextc = V2(0.0f);                 // min_distance at the end point
        bool found[2] = {false, false};      // first interior sample with the opposite sign, per half
        float top[2] = {0.0f, 0.0f};
        const bool want_state = (A.recw == REC_TAIL_STATE);
        const bool any_accept = __ballot((run[0] && EE2.x <= 1.0f) || (run[1] && EE2.y <= 1.0f)) != 0ull;
        bool safe[2] = {true, true};         // FARP: no object's distance can change sign anywhere in this step
        if constexpr (FARP) {
            // |x_q(θ) − x_q| <= δ_q for all θ in [0,1] (integrate_body, "reach bound"), both rays packed
            V2 dl[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                V2 a2 = N::beta[0] * rabs<V2>(k[0][q]);
#pragma unroll
                for (int l = 1; l < 6; l++) a2 = rfma<V2>(V2(N::beta[l]), rabs<V2>(k[l][q]), a2);
                dl[q] = h * rfma<V2>(h, a2, rabs<V2>(u[q]));
            }
            const float guard = 1.0f + 1e-6f;
            for_each_object<float>(A.sc, [&](const DevObject<float>& ob, uint32_t) {
                V2 lhs, rhs;
                if (ob.kind == RTGR_PLANE) {
                    lhs = rabs<V2>(x[0] - V2(ob.p[0]));
                    rhs = rfma<V2>(V2(guard), dl[0], V2(256.0f * eps) * (rabs<V2>(x[0]) + V2(__builtin_fabsf(ob.p[0]))));
                } else if (ob.kind == RTGR_SPHERE) {
                    const V2 X0 = x[1] - V2(ob.p[1]), X1 = x[2] - V2(ob.p[2]), X2 = x[3] - V2(ob.p[3]);
                    const float Rr = ob.p[8];
                    const V2 D0 = rfma<V2>(X0, X0, rfma<V2>(X1, X1, rfma<V2>(X2, X2, V2(-Rr * Rr))));
                    const V2 B = rfma<V2>(dl[1], rfma<V2>(V2(2.0f), rabs<V2>(X0), dl[1]),
                                          rfma<V2>(dl[2], rfma<V2>(V2(2.0f), rabs<V2>(X1), dl[2]), dl[3] * rfma<V2>(V2(2.0f), rabs<V2>(X2), dl[3])));
                    lhs = rabs<V2>(D0);
                    rhs = rfma<V2>(V2(guard), B, V2(256.0f * eps) * (lhs + V2(2.0f * Rr * Rr)));
                } else {
                    V2 px = x[1], py = x[2];
                    asm volatile("" : "+v"(px), "+v"(py));  // keep the disk's root inside this branch
                    lhs = rabs<V2>(V2{disk_distance_fast<float>(ob, px.x, py.x, x[3].x), disk_distance_fast<float>(ob, px.y, py.y, x[3].y)});
                    rhs = V2(guard) * rmax<V2>(dl[3], dl[1] + dl[2]);
                }
                safe[0] = safe[0] && (lhs.x > rhs.x);
                safe[1] = safe[1] && (lhs.y > rhs.y);
            });
        }
        if (!FARP && any_accept) {
            if (want_state) {
                // the caller wants end states: the velocity polynomial u(θ) = u + h Σ_j b_j(θ) k_j of THIS step goes into the
                // slot's record now, whether or not the step turns out to end the ray (a later step overwrites it; the
                // last write is the ending step's) — so that the seven stages are not kept alive across the scan below
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    if (run[hh]) {
                        const RecRef<float> rec{A.hand + idx[hh] * HAND_W, A.rec + idx[hh] * (uint64_t)A.recw};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            rec[REC_U + q] = u[q][hh];
#pragma unroll
                            for (int m = 0; m < 4; m++)
                                rec[REC_CU + 4 * m + q] = h[hh] * rfma<float>(N::r[6][m], k[6][q][hh], rfma<float>(N::r[5][m], k[5][q][hh],
                                    rfma<float>(N::r[4][m], k[4][q][hh], rfma<float>(N::r[3][m], k[3][q][hh], rfma<float>(N::r[2][m], k[2][q][hh],
                                    rfma<float>(N::r[1][m], k[1][q][hh], N::r[0][m] * k[0][q][hh]))))));
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                cc[0][q] = h * u[q];
#pragma unroll
                for (int m = 1; m < 4; m++)
                    cc[m][q] = h2 * rfma<V2>(V2(N::R2[5][m]), k[5][q], rfma<V2>(V2(N::R2[4][m]), k[4][q], rfma<V2>(V2(N::R2[3][m]), k[3][q],
                                    rfma<V2>(V2(N::R2[2][m]), k[2][q], rfma<V2>(V2(N::R2[1][m]), k[1][q], N::R2[0][m] * k[0][q])))));
            }
            // three blocks of three sample points (θ = 1/9…3/9, 4/9…6/9, 7/9, 8/9 and the end point): each object's parameters
            // are fetched once per block, and only three positions of both rays are live at a time
#pragma unroll
            for (int blk = 0; blk < 3; blk++) {
                V2 pos[3][4], dm[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    dm[j] = V2(__builtin_huge_valf());
                    if (blk == 2 && j == 2) {
#pragma unroll
                        for (int q = 0; q < 4; q++) pos[j][q] = xn[q];
                    } else {
                        const V2 th = V2((float)(3 * blk + j + 1) / 9.0f);
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            pos[j][q] = rfma<V2>(th, rfma<V2>(th, rfma<V2>(th, rfma<V2>(th, cc[3][q], cc[2][q]), cc[1][q]), cc[0][q]), x[q]);
                    }
                }
                for_each_object<float>(A.sc, [&](const DevObject<float>& ob, uint32_t) { fold_distances2<3>(ob, pos, dm); });
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (blk == 2 && j == 2) { nextc = dm[j]; continue; }
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const bool hit = (ps[hh] * dm[j][hh] < 0.0f) && !found[hh];
                        top[hh] = hit ? (float)(3 * blk + j + 1) / 9.0f : top[hh];
                        found[hh] = found[hh] || hit;
                    }
                }
            }
        }
        // [budget: controller and records per half]
        // ================= per half: decisions and side effects (what has no packed form) ====================================
        bool commit[2] = {false, false};
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            if (!run[hh]) continue;
            const float EEst2 = EE2[hh];
            const float hs = h[hh], ts = t[hh], pss = ps[hh];
            uint32_t done = 0xffu;
            bool is_event = false, is_interior = false;
            if (EEst2 != EEst2) {
                done = RTGR_RAY_NAN;
            } else {
                // ---- PI controller in log2 space: q = EEst^β1 / qold^β2 / γ
                const float le = 0.5f * flog2(fmax1(EEst2, 1e-37f));
                const float q11 = fexp2(beta1 * le);
                float qf = fexp2(__builtin_fmaf(beta1, le, -beta2 * lq[hh])) * igamma;
                qf = (EEst2 == 0.0f) ? qmax_inv : fclamp1(qf, qmax_inv, qmin_inv);
                if (FARP && EEst2 <= 1.0f && (!safe[hh] || pss == 0.0f)) {
                    // hand the ray, in its PRE-step state, to the NEAR pass (which redoes this step with the full scan)
                    float* hd = A.hand + idx[hh] * HAND_W;
#pragma unroll
                    for (int q = 0; q < 4; q++) { hd[q] = x[q][hh]; hd[4 + q] = u[q][hh]; hd[8 + q] = k[0][q][hh]; }
                    hd[12] = ts; hd[13] = dt[hh]; hd[14] = pss; hd[15] = lq[hh];
                    A.meta[idx[hh] * 3] = nacc[hh]; A.meta[idx[hh] * 3 + 1] = nrej[hh];
                    A.meta[idx[hh] * 3 + 2] = META_HANDED;
                    state[hh] = L_FREE;
                } else if (EEst2 <= 1.0f) {
                    nacc[hh]++;
                    lq[hh] = fmax1(le, lq_init);
                    const float dtnew = dt[hh] * __builtin_amdgcn_rcpf(qf);
                    float tnew = ts + dt[hh];
                    if (rabs(tnew - t1) < 10.0f * eps * rmaxabs<float>(tnew, t1)) tnew = t1;
                    const float nc = nextc[hh];
                    const bool endpoint = !FARP && (pss != 0.0f) && (pss * nc <= 0.0f);
                    const bool interior = !FARP && found[hh] && (pss != 0.0f) && !endpoint;
                    if (endpoint || interior) {
                        top[hh] = endpoint ? 1.0f : top[hh];
                        is_event = true;
                        is_interior = interior;
                        done = RTGR_RAY_EVENT;
                    }
                    if (!is_event) {
                        if constexpr (!FARP) ps[hh] = rsign(nc);   // (FAR: proven above that no distance changes sign in this step)
                        commit[hh] = true;
                        t[hh] = tnew;
                        dt[hh] = fmin1(dtmax, dtnew);
                        if (!(tnew < t1)) done = RTGR_RAY_LAMBDA1;
                        else if (nacc[hh] + nrej[hh] >= A.opt.max_steps) done = RTGR_RAY_MAXSTEPS;
                        else if (!(tnew + dt[hh] > tnew)) done = RTGR_RAY_DTMIN;
                    }
                } else {
                    nrej[hh]++;
                    dt[hh] = dt[hh] * __builtin_amdgcn_rcpf(fmin1(qmin_inv, q11 * igamma));
                    if (nacc[hh] + nrej[hh] >= A.opt.max_steps) done = RTGR_RAY_MAXSTEPS;
                    else if (!(ts + dt[hh] > ts)) done = RTGR_RAY_DTMIN;
                }
            }
            if (done != 0xffu) {
                const RecRef<float> rec{A.hand + idx[hh] * HAND_W, A.rec + idx[hh] * (uint64_t)A.recw};
                if (is_event) {   // the step's position polynomial, from the step-START state
#pragma unroll
                    for (int q = 0; q < 4; q++) rec[REC_X + q] = x[q][hh];
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int q = 0; q < 4; q++) rec[REC_C + 4 * m + q] = cc[m][q][hh];
                    rec[REC_PS] = pss;
                    rec[REC_TOP] = top[hh];
                    rec[REC_T] = ts;
                    rec[REC_H] = hs;
                    // (REC_U / REC_CU of this step were written before the scan when the caller wants end states)
                } else {   // ended without an event (λ1, step cap, dt underflow, NaN): the state as it stands, θ = 0
#pragma unroll
                    for (int q = 0; q < 4; q++) rec[REC_X + q] = commit[hh] ? xn[q][hh] : x[q][hh];
                    rec[REC_PS] = 0.0f;
                    rec[REC_TOP] = 0.0f;
                    rec[REC_T] = t[hh];
                    rec[REC_H] = 0.0f;
                    if (want_state) {
#pragma unroll
                        for (int q = 0; q < 4; q++) rec[REC_U + q] = commit[hh] ? un[q][hh] : u[q][hh];
                    }
                }
                uint32_t* mt = A.meta + idx[hh] * 3;
                mt[0] = nacc[hh];
                mt[1] = nrej[hh];
                mt[2] = done | (is_interior ? 0x100u : 0u);
                c_rays += 1; c_acc += nacc[hh]; c_rej += nrej[hh];
                c_ev += is_event; c_int += is_interior; c_nf += (done >= RTGR_RAY_MAXSTEPS);
                state[hh] = L_FREE;
            }
        }
        // [budget: commit]
        // ---- commit the accepted steps: per half (the only place besides the refill where the ray states are assigned) -------
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            if (commit[hh]) {
#pragma unroll
                for (int q = 0; q < 4; q++) { x[q][hh] = xn[q][hh]; u[q][hh] = un[q][hh]; k0[q][hh] = k[6][q][hh]; }
            }
        }
        // [budget: end]
    }
    if (A.counters) {
        const unsigned long long s0 = wave_sum(c_rays), s1 = wave_sum(c_acc), s2 = wave_sum(c_rej),
                                 s4 = wave_sum(c_ev), s5 = wave_sum(c_int), s6 = wave_sum(c_nf);
        if (lane == 0) {
            atomicAdd(&A.counters[0], s0);
            atomicAdd(&A.counters[1], s1);
            atomicAdd(&A.counters[2], s2);
            atomicAdd(&A.counters[3], 6ull * (s1 + s2) + 2ull * s0);  // RHS evaluations: 6 per attempt + 2 per ray
            atomicAdd(&A.counters[4], s4);
            atomicAdd(&A.counters[5], s5);
            atomicAdd(&A.counters[6], s6);
        }
    }
}

#ifndef RTGR_WAVES_PER_SIMD_PACKED
#define RTGR_WAVES_PER_SIMD_PACKED 2   // two rays per lane: twice the state of the scalar kernel (151 registers) — two waves = 256 rays per SIMD
#endif
template <int METRIC, bool SPIN>
__global__ __launch_bounds__(64, RTGR_WAVES_PER_SIMD_PACKED) void integrate2_kernel(const IntegrateArgs<float> A) {
    integrate2_body<METRIC, SPIN>(A);
}
#ifndef RTGR_WAVES_PER_SIMD_PACKED_FAR
#define RTGR_WAVES_PER_SIMD_PACKED_FAR 3   // the scan-free instantiation: 159-168 registers
#endif
template <int METRIC, bool SPIN>
__global__ __launch_bounds__(64, RTGR_WAVES_PER_SIMD_PACKED_FAR) void integrate2_far_kernel(const IntegrateArgs<float> A) {
    integrate2_body<METRIC, SPIN, true>(A);
}

}  // namespace rtgr
