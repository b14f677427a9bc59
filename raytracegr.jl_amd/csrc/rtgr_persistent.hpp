// rtgr_persistent.hpp — the production trace pipeline:
//     prepare (camera ray, queue key, u̇(y0), initial dt) -> [order scan/scatter] -> integrate<FAR> -> integrate<NEAR>
//     -> resolve                                                              (rtgr_pipeline.hpp: launch_trace)
//
// Why a pipeline instead of one loop per ray (the simple tile kernel, trace_kernel in rtgr_pipeline.hpp, keeps that shape):
//   * rays need 27…991 Tsit5 step attempts inside one image; a wave that owns 64 fixed rays idles until
//     its longest ray ends, and each ray's end-of-life episode (bracketed root-find on the dense output, colouring,
//     next ray's camera set-up) would run with one lane active;
//   * so the integrate kernel treats a wave as a pool of 64 ray slots: a lane whose ray has ended writes a small
//     EVENT RECORD (the position polynomial of its last step) to HBM, and is re-filled from a global ray queue on the
//     next iteration — a wavefront ballot finds the free lanes, which take ids from the wave's slice of the queue;
//   * everything that happens once per ray and diverges — root-finding, the colouring rule with its acos/atan2 —
//     runs afterwards in the resolve kernel with every lane busy; the camera (make_canvas) runs before, likewise;
//   * the ContinuousCallback scan (8 interior samples per accepted step) is replaced, while a ray is out of reach of
//     every object, by a rigorous bound (FAR pass); rays within reach are handed to the NEAR pass (full scan);
//   * the queue is ordered longest-expected-first so that few rays per lane still balance (order_* kernels).
//   HBM traffic of the hand-offs: ≈ 1 kB per ray, against ≈ 0.23 Mflop of integration per ray.
//
// Step body (per lane, registers only):
//   * Nyström form: since ẋ = u (src/RayTraceGR.jl:360) stage positions are x + h c_s u + h² Σ A2[s][l] k_l — only the
//     seven acceleration 4-vectors k_l are stored (28 scalars), not the 7 x 8 stage derivatives;
//   * a fresh ray's u̇(y0) and Hairer initial dt (2 RHS evaluations) are computed by prepare_kernel, one thread per
//     ray, into the same 16-scalar record the FAR pass uses to hand a ray over — the integrate loop has ONE way to
//     start a ray and no per-ray set-up code of its own (an in-loop "init pseudo-step" ran its ~200 extra
//     instructions for the whole wave in the ~26 % of iterations in which some lane had just been refilled);
//   * error norm, PI controller and initial-dt formula run in f32 on the (otherwise idle) f32 VALU/transcendental
//     path; they only steer the step size — the state arithmetic is all R (f64);
//   * ContinuousCallback (NEAR / FULL): the position interpolant is put in polynomial form once per accepted step and
//     evaluated at the 8 interior sample points θ = j/9; distances are evaluated object-major so
//     each object's parameters are fetched once per block of sample points.
#pragma once
#include "rtgr_integrator.hpp"
#include "rtgr_tsit5_tables.hpp"

#ifndef RTGR_GHOST_LANES
#define RTGR_GHOST_LANES 1  // lanes without a ray run the step's control block too (few-lane VALU instructions are 2.4 x dearer)
#endif
#ifndef RTGR_ROOT_SHORTCUT
#define RTGR_ROOT_SHORTCUT 1
#endif
#ifndef RTGR_WAVES_PER_SIMD_GENERIC
#define RTGR_WAVES_PER_SIMD_GENERIC 2  // generic dual-number RHS: ~270 registers wanted; 2 waves with a small spill beat 1 wave (measured 6.72 vs 6.22 Gstep/s)
#endif
#ifndef RTGR_LDSK_GENERIC
#define RTGR_LDSK_GENERIC 0   // generic RHS kernels keep k[1..5] in LDS (experiment: see DESIGN §4.2 "LDS stage storage")
#endif
#ifndef RTGR_LDSK_SPIN_FAR
#define RTGR_LDSK_SPIN_FAR 0  // the a != 0 FAR pass keeps k[1..5] in LDS (experiment, with RTGR_WAVES_PER_SIMD_SPIN_FAR=4)
#endif
#ifndef RTGR_WAVES_PER_SIMD_SPIN_FAR
#define RTGR_WAVES_PER_SIMD_SPIN_FAR RTGR_WAVES_PER_SIMD_FAR
#endif
#ifndef RTGR_WAVES_PER_SIMD_GENERIC_F32
#define RTGR_WAVES_PER_SIMD_GENERIC_F32 3  // generic dual-number RHS in Float32: half the register bytes of the f64 kernel
#endif
#ifndef RTGR_WAVES_PER_SIMD_FAR
#define RTGR_WAVES_PER_SIMD_FAR 3  // the FAR pass has no sample-point arrays: <=168 registers, three waves per SIMD
#endif
#ifndef RTGR_WAVES_PER_SIMD_F32
#define RTGR_WAVES_PER_SIMD_F32 3  // Float32 NEAR / FULL passes: 146-158 registers, three waves fit (2048²: 4.02 -> 3.50 ms; four: 3.82)
#endif
#ifndef RTGR_WAVES_PER_SIMD
#define RTGR_WAVES_PER_SIMD 2  // 2 -> <=256 VGPR+AGPR per lane; 1 -> the whole 512-entry file
#endif

namespace rtgr {

// fast f32 helpers for the step-size machinery
// (uniform_(), rtgr_physics.hpp: a wave-uniform value computed with vector instructions — there is no scalar f64 ALU —
//  lives in a VGPR and, in a kernel squeezed to 127 registers, gets spilled to scratch and reloaded in the loop; through
//  readfirstlane it lives in SGPRs.)
// one-instruction f32 max / min / clamp (fmaxf / fminf spend a second v_max x, x on canonicalising each computed operand)
RTGR_DEV float fmax1(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTGR_DEV float fmin1(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RTGR_DEV float fclamp1(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }  // v_med3_f32
RTGR_DEV float flog2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
RTGR_DEV float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32

typedef float float2_t __attribute__((ext_vector_type(2)));

// lanes below `lane` set in mask
RTGR_DEV uint32_t mask_rank(unsigned long long mask, uint32_t lane) {
    return (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
}

// … of a SPHERE (the leading objects of the regrouped list, DevScene: no dispatch on the kind)
template <class R, int P>
RTGR_DEV void fold_sphere(const DevObject<R>& o, const R (&pos)[P][4], R (&dmin)[P]) {
    const R cx = o.p[1], cy = o.p[2], cz = o.p[3], Rr = o.p[8];
    const R nR2 = -Rr * Rr;
    if (Rr < R(0)) {                                                                   // :415-419, sign(R) * (|x − c|² − R²)
#pragma unroll
        for (int p = 0; p < P; p++) {
            const R dx = pos[p][1] - cx, dy = pos[p][2] - cy, dz = pos[p][3] - cz;
            dmin[p] = rmin(dmin[p], -rfma(dx, dx, rfma(dy, dy, rfma(dz, dz, nR2))));
        }
    } else {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const R dx = pos[p][1] - cx, dy = pos[p][2] - cy, dz = pos[p][3] - cz;
            dmin[p] = rmin(dmin[p], rfma(dx, dx, rfma(dy, dy, rfma(dz, dz, nR2))));
        }
    }
}

// distances of one object at P sample positions folded into dmin[] (object-major: parameters fetched once)
template <class R, int P>
RTGR_DEV void fold_distances(const DevObject<R>& o, const R (&pos)[P][4], R (&dmin)[P]) {
    if (o.kind == RTGR_PLANE) {                                                        // src/RayTraceGR.jl:399-401
        const R tm = o.p[0];
#pragma unroll
        for (int p = 0; p < P; p++) dmin[p] = rmin(dmin[p], pos[p][0] - tm);
    } else if (o.kind == RTGR_SPHERE) {                                                // :415-419
        const R cx = o.p[1], cy = o.p[2], cz = o.p[3], Rr = o.p[8];
        const R nR2 = -Rr * Rr;
        if (Rr < R(0)) {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const R dx = pos[p][1] - cx, dy = pos[p][2] - cy, dz = pos[p][3] - cz;
                dmin[p] = rmin(dmin[p], -rfma(dx, dx, rfma(dy, dy, rfma(dz, dz, nR2))));
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const R dx = pos[p][1] - cx, dy = pos[p][2] - cy, dz = pos[p][3] - cz;
                dmin[p] = rmin(dmin[p], rfma(dx, dx, rfma(dy, dy, rfma(dz, dz, nR2))));
            }
        }
#ifdef RTGR_USER_OBJECTS
    } else if (o.kind == RTGR_USER_OBJECT) {                                           // the unit's own distance method (:377-386)
#pragma unroll
        for (int p = 0; p < P; p++) dmin[p] = rmin(dmin[p], rtgr_user_distance<R>(o.type, pos[p], o.p));
#endif
    } else {
        // RTGR_DISK: the scan needs the distance's SIGN only — disk_sign_distance reads it off x² + y² without a square
        // root, exactly (rtgr_physics.hpp).  The asm barrier pins the operands inside this branch: without it LLVM hoists
        // the (loop-invariant) x² + y² of every sample point out of the object loop, for scenes that contain no disk at all.
#pragma unroll
        for (int p = 0; p < P; p++) {
            R px = pos[p][1], py = pos[p][2];
            asm volatile("" : "+v"(px), "+v"(py));
            dmin[p] = rmin(dmin[p], disk_sign_distance<R>(o, px, py, pos[p][3]));
        }
    }
}

// FAR pass: append the wave's collected early-list entries (LDS) to the global list; ONE atomic for all of them
constexpr uint32_t RTGR_EARLY_BUF = 128;
template <class R>
RTGR_DEV void flush_early(const IntegrateArgs<R>& A, const uint32_t* buf, uint32_t cnt, uint32_t lane) {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(A.ctrl + 6, (unsigned long long)cnt);
    base = uniform64(base);
    for (uint32_t i = lane; i < cnt; i += 64) A.early[base + i] = buf[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// integrate kernel
// ---------------------------------------------------------------------------------------------------------------------
// The kernel body is a device function so that run-time generated units (user metrics, rtgr_user_unit.hip.in) can
// wrap it in extern "C" kernels of their own.
template <class R, int METRIC, bool SPIN, bool NPTS10, int MODE, bool LDSK = false>
RTGR_DEV void integrate_body(const IntegrateArgs<R>& A) {
    using N = Tsit5N<R>;
    const uint32_t lane = threadIdx.x & 63;
    // NEAR visits the early list first (ids [0, n_early) of its queue), then every ray id, picking up the rays flagged
    // for it that are not on the list.  The early rays then run packed together from the start of the pass — all far from
    // every object most of the time, so their waves skip the scan — instead of sitting one or two to a wave of
    // short-lived rays and keeping it alive when the queue is empty (measured before: the median wave ended at 2.2 ms,
    // the pass at 3.3 ms).
    uint64_t n_early = 0;
    if (MODE == MODE_NEAR && A.early) {  // (wave-uniform for the compiler too: it sets the loop's exit)
        const unsigned long long c = A.ctrl[6];
        n_early = uniform64(c);
    }
    const uint64_t total = A.n + n_early;
    unsigned long long* const queue = (MODE == MODE_NEAR) ? A.ctrl + 1 : A.ctrl;
    const MetricK<R> MK = metric_consts<R>(A.sc.M, A.sc.a);
    const R reltol = A.opt.reltol, abstol = A.opt.abstol;
    const R t0 = A.opt.lambda0, t1 = A.opt.lambda1, dtmax = uniform_(A.opt.lambda1 - A.opt.lambda0);
    const float igamma = 1.0f / 0.9f, qmin_inv = 5.0f, qmax_inv = 0.1f;
    const float lq_init = -13.287712379549449f;  // log2(qoldinit = 1e-4)
    const float beta1 = 0.14f, beta2 = 0.08f;
    const int npts = (int)A.opt.interp_points;
    const R eps = sizeof(R) == 8 ? R(2.220446049250313e-16) : R(1.1920929e-7);
    const R t1_snap = uniform_(t1 - R(16) * eps * rmaxabs<R>(t0, t1));   // below this λ the snap-to-λ1 test cannot hold
    // (the NEAR pass's scan mask, below: one bit per 2^mask_shift neighbours of the device list — one per object up to 64 objects)
    const uint32_t mask_shift = A.sc.nobj > 64u ? 32u - (uint32_t)__builtin_clz((A.sc.nobj - 1u) >> 6) : 0u;

    int state = L_FREE;
    bool exhausted = false;
    R x[4], u[4], k0[4];  // loop-carried ray state: position, velocity, FSAL acceleration u̇(x, u)
    R t = t0, dt = R(0), ps = R(0);
    float lq = lq_init;
    uint64_t idx = 0;
    uint32_t nacc = 0, nrej = 0, nacc0 = 0, c_maxnear = 0, safe_streak = 0;
    uint32_t c_rays = 0, c_acc = 0, c_rej = 0, c_ev = 0, c_int = 0, c_nf = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { x[q] = R(1); u[q] = R(0); k0[q] = R(0); }

    // wave-local slice of the global ray queue: ids [q_next, q_end) were popped with ONE atomic and are dealt to lanes as
    // they free up.  (One device-scope atomic per ray on a single word saturates at ~90 M/s chip-wide — measured: it
    // stalled the FAR pass 5x — so waves pop up to RTGR_QUEUE_CHUNK ids at a time; fewer when the whole job is only a few
    // hundred rays per wave, or the last chunks would unbalance the waves.)
    uint64_t q_next = 0, q_end = 0;
    bool first_pop = true;
    bool first_early = true, early_done = false;  // NEAR: the early list's own cursor (ctrl[7])
    __shared__ uint32_t early_buf[RTGR_EARLY_BUF];  // one wave per workgroup: private to the wave (used by the FAR pass)
    __shared__ R ldsk_mem[LDSK ? 20 * 64 : 1];
    volatile R* const ldsk = ldsk_mem + lane;   // (volatile: re-read at each use instead of being kept live in registers)
    uint32_t e_cnt = 0;                             // entries in early_buf (wave-uniform)
#ifdef RTGR_ROOT_STATS
    const unsigned long long dbg_t0 = wall_clock64();
    unsigned long long dbg_iters = 0, dbg_rays = 0;
#endif
    const unsigned long long qchunk = A.queue_chunk;
    // Wave ages.  A SIMD's arbiter issues from its OLDEST ready wave first, and a persistent grid never renews its
    // waves: measured (round 1; its successor is tools/wave_timeline.py on a -DRTGR_ROOT_STATS build) 2.9 us per iteration for the first-launched
    // third of the FAR pass's workgroups, 5-7 us for the second, 14 us for the youngest (40 us on the 4-wave grid).
    // Throughput does not care, the tail of a small launch does: a young wave that holds rays of 950 steps needs
    // the whole pass for them.  So (a) the head of the longest-first queue goes to the low workgroup indices (first
    // pop by index), and (b) in small launches (fair_shift != 0) the waves of a SIMD take turns at the top priority
    // (s_setprio, time slices of 2^fair_shift clocks): per-iteration times 4.1-7.3 us instead of 2.9-14.5, FAR pass
    // 7.08 -> 6.31 ms at 1 M rays; no effect at 2-4 M rays, 1 % SLOWER at 16.8 M (where there is no tail to win and the
    // rotation only disturbs the arbiter) and 15 % slower at <= 0.5 M (one or two rays per lane: the old waves should
    // race through the longest ones), hence by launch size.  Serving the queue from both ends — old
    // waves from the head, young ones from the tail — was worse at every size: the tail-servers then run
    // shortest-first, the worst order inside a wave.
    const uint32_t n_simd = A.n_simd ? A.n_simd : 1024u;              // SIMDs of the device (MI355X: 256 CUs x 4)
    const uint32_t n_cls = (gridDim.x + n_simd - 1u) / n_simd;        // waves per SIMD of this launch
    const uint32_t my_cls = blockIdx.x / n_simd;
    const bool fair = A.fair_shift != 0u && n_cls > 1u;
    for (;;) {
        if (fair) {
            const uint32_t turn = (my_cls + (uint32_t)(__builtin_readcyclecounter() >> A.fair_shift)) % n_cls;
            switch (turn) {  // s_setprio takes an immediate
                case 0: __builtin_amdgcn_s_setprio(3); break;
                case 1: __builtin_amdgcn_s_setprio(2); break;
                case 2: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
        }
        // [budget: refill]
        // ================= refill: free lanes take ray ids from the wave's slice of the queue ====================
        const bool resume = (MODE == MODE_NEAR || A.pick_flag != 0u);
        unsigned long long m_need = __ballot(state == L_FREE);
        while (m_need != 0ull) {
            if (q_next == q_end) {
                if (exhausted) break;
                // The FIRST pop takes exactly the ids the wave can start right away: in longest-first order the head of
                // the queue holds the long rays, and ids parked in a wave's slice would only start when one of its
                // (equally long) first rays ends — simulated makespan 1978 vs 1213 steps at 5 rays per lane.
                // ... and it takes them by WAVE INDEX, not by atomic order: queue positions [64 b, 64 b + 64) go to
                // workgroup b, the atomic head serves positions from 64 * gridDim.x on (see the note on wave ages above).
                const unsigned long long first_span = 64ull * gridDim.x;
                if (MODE == MODE_NEAR && !early_done) {
                    // the early list has a head of its own and goes out 64 entries at a time: they are long-stayers, and
                    // a 256-entry chunk would be four rounds of them one after the other in the same wave (measured:
                    // the late waves of the pass were the ones that had popped such a chunk)
                    // (… unless the list is most of the job — a distant camera, every ray shorter than near_early steps:
                    //  then it is not a list of exceptions any more and goes out in full queue chunks, or its 64-id pops
                    //  are 260 k atomics at 4096², 3 ms: tools/scheduler_check.py, camera far20)
                    const unsigned long long echunk = (n_early * 8ull > A.n) ? qchunk : 64ull;
                    unsigned long long eb = 0;
                    if (first_early) eb = echunk * blockIdx.x;
                    else if (lane == 0) eb = atomicAdd(A.ctrl + 7, echunk) + echunk * gridDim.x;
                    first_early = false;
                    eb = uniform64(eb);
                    if (eb >= n_early) { early_done = true; continue; }
                    q_next = eb;
                    q_end = (eb + echunk) < n_early ? (eb + echunk) : n_early;
                    continue;
                }
                const unsigned long long amount = first_pop ? 64ull : qchunk;
                unsigned long long base = 0;
                if (first_pop) base = 64ull * blockIdx.x;
                else if (lane == 0) base = atomicAdd(queue, amount) + first_span;
                first_pop = false;
                // wave-uniform BY CONSTRUCTION and, through readfirstlane, also for the compiler: q_next / q_end /
                // exhausted then live in SGPRs and the loop's exits are scalar branches.  (With a __shfl the compiler
                // had to treat the loop exit as divergent and copied all 12 loop-carried f64 state registers to
                // shadow registers and back at the latch: 41 v_mov per iteration, 4.7 % of the kernel's instructions.)
                base = uniform64(base);
                base += n_early;  // positions [0, n_early) are the early list's
                q_next = base < total ? base : total;
                q_end = (base + amount) < total ? (base + amount) : total;
                if (base + amount >= total) exhausted = true;
                if (q_next == q_end) break;
            }
            const uint64_t avail = q_end - q_next;
            const uint32_t rank = mask_rank(m_need, lane);
            if (state == L_FREE && rank < avail) {
                const uint64_t w = q_next + rank;
                // A resuming pass (NEAR, hand-back rounds) visits every ray id and picks up the rays flagged for it.
                uint64_t id = resume ? w : (A.order ? (uint64_t)A.order[w] : w);
                bool take = true;
                if (resume) {
                    const bool listed = MODE == MODE_NEAR && w < n_early;
                    if (MODE == MODE_NEAR) id = listed ? (uint64_t)A.early[w] : w - n_early;
                    // ONE word decides: which sweep a ray belongs to is part of its flag.  (Deciding it from the step count
                    // in meta[0] raced with the wave that finishes the ray and overwrites meta[0..2] word by word: 3 rays in
                    // 4 M were traced twice — found by tools/stress_determinism.py.)
                    take = A.meta[id * 3 + 2] == (listed ? META_HANDED_EARLY : A.pick_flag);
                }
                if (take) {
                    idx = id;
                    state = L_TAKEN;
                }
            }
            const uint32_t cnt = (uint32_t)__builtin_popcountll(m_need);
            q_next += (cnt < avail) ? cnt : avail;
            m_need = __ballot(state == L_FREE);
        }
        // Every ray starts from a 16-scalar record {x, u, u̇, t, dt, sign(min_distance), log2 q_old}: written by
        // prepare_kernel for a fresh ray (u̇(y0) and the Hairer initial dt), by the FAR pass for a ray it hands over.
        // (The loads, like the commit at the end of the step, are a plain conditional assignment at the top level of the
        // loop body: assigned inside nested divergent regions, the 12 f64 state registers were shadow-copied at every
        // region entry, at the loop latch and at the loop header — 85 v_mov per iteration.)
#ifdef RTGR_ROOT_STATS
        dbg_iters++;
        dbg_rays += __builtin_popcountll(__ballot(state == L_TAKEN));
#endif
        if (state == L_TAKEN) {
            const R* hd = A.hand + idx * HAND_W;
#pragma unroll
            for (int q = 0; q < 4; q++) { x[q] = hd[q]; u[q] = hd[4 + q]; k0[q] = hd[8 + q]; }
            t = hd[12]; dt = hd[13]; ps = hd[14]; lq = (float)hd[15];
            nacc = resume ? A.meta[idx * 3] : 0u;
            nrej = resume ? A.meta[idx * 3 + 1] : 0u;
            nacc0 = nacc;
            safe_streak = 0;
            state = L_RUN;
        }
        if (exhausted && q_next == q_end && state == L_FREE) state = L_EXIT;
        // The refill loop above only gives up when the queue is exhausted, so a wave without a running lane is done.
        // (One exit and one back-edge: a `continue` here was a second latch path and cost a round trip of phi copies.)
        if (__ballot(state == L_RUN) == 0ull) break;

        // [budget: stage sums]
        // ================= one Tsit5 attempt per runnable lane ====================================================
        // The seven stages run in UNIFORM control flow: a lane without a ray computes along with h = 0 on whatever
        // state it holds (a VALU instruction costs the same with 1 or 64 lanes active) and nothing it computes is
        // used.  Only the decisions and side effects below are per-lane.
        const bool run = (state == L_RUN);
        bool commit = false;
        bool list_it = false;  // FAR: this lane's ray goes on the early list (set at hand-over)
        R xn[4], un[4], k[7][4];  // k[l] = acceleration at stage l+1 (k[0] is the FSAL slot, k[6] the next one)
        // LDSK (SURVEY §7.4 "k-stage storage in LDS if VGPR > 256"; BASELINE config 4's "LDS-tiled variant"): the stage
        // accelerations k[1..5] — 20 scalars, 40 VGPRs in f64 — live in LDS between the stage that produces them and the
        // stage sums / error estimate / reach bound that read them; slot (l, q) of lane i is word ((l−1)·4 + q)·64 + i, so
        // every access of a wave is 64 consecutive scalars (conflict-free ds_read/write_b64)
#define KL(l, q) ((LDSK && (l) >= 1 && (l) <= 5) ? (R)ldsk[(((l) - 1) * 4 + (q)) * 64] : k[l][q])
#define KSTORE(l)                                                                   \
        if constexpr (LDSK) {                                                       \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) ldsk[(((l) - 1) * 4 + q_) * 64] = k[l][q_]; \
        }
#pragma unroll
        for (int q = 0; q < 4; q++) k[0][q] = k0[q];
        {
            if (run) dt = rmin(dt, t1 - t);
            const R h = run ? dt : R(0);
            const R h2 = h * h;
            R X[3], U[4];
            // x^t at the stage: formed only for a user metric that was not declared stationary (such a metric was
            // traced frozen at t = 0 — the reference evaluates christoffel(metric, x) at the full 4-position, :358-363)
            constexpr bool TDEP = needs_stage_time<METRIC>();
            R Xt = R(0);
            // ---- stage 2 -------------------------------------------------------------------------------------
            {
                const R ha = h * N::a[1][0], hc = h * N::c[1];
#pragma unroll
                for (int q = 0; q < 4; q++) U[q] = rfma(ha, k[0][q], u[q]);
#pragma unroll
                for (int q = 0; q < 3; q++) X[q] = rfma(hc, u[1 + q], x[1 + q]);
                if constexpr (TDEP) Xt = rfma(hc, u[0], x[0]);
            }
            // [budget: rhs]
            accel<R, METRIC, SPIN, true>(X, U, MK, k[1], Xt);
            // [budget: stage sums]
            KSTORE(1);
            // ---- stage 3 -------------------------------------------------------------------------------------
            {
                const R w1 = h * N::a[2][1], w0 = h * N::a[2][0], hc = h * N::c[2], h2a = h2 * N::A2[2][0];
#pragma unroll
                for (int q = 0; q < 4; q++) U[q] = rfma(w1, KL(1, q), rfma(w0, k[0][q], u[q]));
#pragma unroll
                for (int q = 0; q < 3; q++) X[q] = rfma(h2a, k[0][1 + q], rfma(hc, u[1 + q], x[1 + q]));
                if constexpr (TDEP) Xt = rfma(h2a, k[0][0], rfma(hc, u[0], x[0]));
            }
            // [budget: rhs]
            accel<R, METRIC, SPIN, true>(X, U, MK, k[2], Xt);
            // [budget: stage sums]
            KSTORE(2);
            // ---- stages 4, 5, 6 ----------------------------------------------------------------------------------
#pragma unroll
            for (int q = 0; q < 4; q++)
                U[q] = rfma(h, rfma(N::a[3][2], KL(2, q), rfma(N::a[3][1], KL(1, q), N::a[3][0] * k[0][q])), u[q]);
#pragma unroll
            for (int q = 0; q < 3; q++)
                X[q] = rfma(h2, rfma(N::A2[3][1], KL(1, 1 + q), N::A2[3][0] * k[0][1 + q]),
                            rfma(h * N::c[3], u[1 + q], x[1 + q]));
            if constexpr (TDEP) Xt = rfma(h2, rfma(N::A2[3][1], KL(1, 0), N::A2[3][0] * k[0][0]), rfma(h * N::c[3], u[0], x[0]));
            // [budget: rhs]
            accel<R, METRIC, SPIN, true>(X, U, MK, k[3], Xt);
            // [budget: stage sums]
            KSTORE(3);
#pragma unroll
            for (int q = 0; q < 4; q++)
                U[q] = rfma(h, rfma(N::a[4][3], KL(3, q), rfma(N::a[4][2], KL(2, q), rfma(N::a[4][1], KL(1, q),
                            N::a[4][0] * k[0][q]))), u[q]);
#pragma unroll
            for (int q = 0; q < 3; q++)
                X[q] = rfma(h2, rfma(N::A2[4][2], KL(2, 1 + q), rfma(N::A2[4][1], KL(1, 1 + q), N::A2[4][0] * k[0][1 + q])),
                            rfma(h * N::c[4], u[1 + q], x[1 + q]));
            if constexpr (TDEP) Xt = rfma(h2, rfma(N::A2[4][2], KL(2, 0), rfma(N::A2[4][1], KL(1, 0), N::A2[4][0] * k[0][0])),
                                          rfma(h * N::c[4], u[0], x[0]));
            // [budget: rhs]
            accel<R, METRIC, SPIN, true>(X, U, MK, k[4], Xt);
            // [budget: stage sums]
            KSTORE(4);
#pragma unroll
            for (int q = 0; q < 4; q++)
                U[q] = rfma(h, rfma(N::a[5][4], KL(4, q), rfma(N::a[5][3], KL(3, q), rfma(N::a[5][2], KL(2, q),
                            rfma(N::a[5][1], KL(1, q), N::a[5][0] * k[0][q])))), u[q]);
#pragma unroll
            for (int q = 0; q < 3; q++)
                X[q] = rfma(h2, rfma(N::A2[5][3], KL(3, 1 + q), rfma(N::A2[5][2], KL(2, 1 + q), rfma(N::A2[5][1], KL(1, 1 + q),
                            N::A2[5][0] * k[0][1 + q]))), rfma(h * N::c[5], u[1 + q], x[1 + q]));
            if constexpr (TDEP) Xt = rfma(h2, rfma(N::A2[5][3], KL(3, 0), rfma(N::A2[5][2], KL(2, 0), rfma(N::A2[5][1], KL(1, 0),
                                          N::A2[5][0] * k[0][0]))), rfma(h * N::c[5], u[0], x[0]));
            // [budget: rhs]
            accel<R, METRIC, SPIN, true>(X, U, MK, k[5], Xt);
            // [budget: stage sums]
            KSTORE(5);
            // ---- stage 7 = the step result (FSAL) ----------------------------------------------------------------
#pragma unroll
            for (int q = 0; q < 4; q++) {
                un[q] = rfma(h, rfma(N::a[6][5], KL(5, q), rfma(N::a[6][4], KL(4, q), rfma(N::a[6][3], KL(3, q),
                             rfma(N::a[6][2], KL(2, q), rfma(N::a[6][1], KL(1, q), N::a[6][0] * k[0][q]))))), u[q]);
                xn[q] = rfma(h2, rfma(N::A2[6][4], KL(4, q), rfma(N::A2[6][3], KL(3, q), rfma(N::A2[6][2], KL(2, q),
                             rfma(N::A2[6][1], KL(1, q), N::A2[6][0] * k[0][q])))), rfma(h * N::c[6], u[q], x[q]));
            }
            // [budget: rhs]
            accel<R, METRIC, SPIN, true>(xn + 1, un, MK, k[6], xn[0]);

            // The error estimate, the controller, the reach bound and (NEAR) the sample-point scan ALSO run in uniform control
            // flow: a wave64 VALU instruction with few lanes enabled is not cheaper on this chip, it is 2.4 x DEARER — f64
            // arithmetic with <= 8 lanes in EXEC issues at 12.4 instead of 5.25 clocks, f32 with <= 16 lanes likewise
            // (tools/micro/exec_mask_rates.hip, profiles/r03/exec_mask_rates.log) — and the waves that decide when a small
            // launch ends (the last rays of the FAR pass, the long stayers of the NEAR pass) are exactly the ones with a
            // handful of live lanes.  So lanes without a ray ("ghosts") walk through this block too, on whatever state they
            // hold and h = 0; every side effect below is gated by `run`, and their decisions are wiped at the end of the block.
            // (-DRTGR_GHOST_LANES=0: the block under the running lanes' EXEC mask, for A/B.)
            if (RTGR_GHOST_LANES ? true : run) {
        // [budget: error norm]
                // ---- embedded error, residual norm in f32 ----------------------------------------
                // ũ_q = h e_q with e = (Σ b̃_l k_l, h Σ BT2_l k_l + Σb̃ u): the common factor h is applied ONCE, to the f32 sum
                // of squares (eight f64 products less per step), and the (u, x) pair of a component goes through the
                // residual arithmetic as one packed f32 operand (v_pk_mul_f32 / v_pk_fma_f32).
                float2_t acc2 = {0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const R eu = rfma(N::bt[6], k[6][q], rfma(N::bt[5], KL(5, q), rfma(N::bt[4], KL(4, q),
                                     rfma(N::bt[3], KL(3, q), rfma(N::bt[2], KL(2, q), rfma(N::bt[1], KL(1, q), N::bt[0] * k[0][q]))))));
                    const R ex = rfma(h, rfma(N::BT2[5], KL(5, q), rfma(N::BT2[4], KL(4, q), rfma(N::BT2[3], KL(3, q),
                                     rfma(N::BT2[2], KL(2, q), rfma(N::BT2[1], KL(1, q), N::BT2[0] * k[0][q]))))), N::sbt * u[q]);
                    const float2_t isk = {__builtin_amdgcn_rcpf((float)rfma(rmaxabs<R>(u[q], un[q]), reltol, abstol)),
                                          __builtin_amdgcn_rcpf((float)rfma(rmaxabs<R>(x[q], xn[q]), reltol, abstol))};
                    const float2_t e = {(float)eu, (float)ex};
                    const float2_t r2 = e * isk;
                    acc2 = __builtin_elementwise_fma(r2, r2, acc2);
                }
                // EEst² — the square root is never taken: the tests are EEst <= 1 <=> EEst² <= 1, and the controller works with
                // log2 EEst = ½ log2 EEst² (sqrtf's IEEE expansion was ~15 instructions per step)
                // (the two factors of h one after the other, not as h²: (float)h squared underflows below |h| ~ 1e-19 while the sum may
                //  already be huge — 0 · inf = NaN would end the ray as RTGR_RAY_NAN instead of DTMIN, a flushed h² would ACCEPT the
                //  step with maximum growth; same three multiplies.  ADVICE r3.)
                const float hf32 = (float)h;
#ifdef RTGR_OLD_ERRNORM
                const float EEst2 = (acc2.x + acc2.y) * (0.125f * hf32 * hf32);
#else
                const float EEst2 = ((acc2.x + acc2.y) * hf32) * (0.125f * hf32);
#endif
        // [budget: controller]
                uint32_t done = 0xffu;  // 0xff = still running, else rtgr_ray_status
                bool is_event = false, is_interior = false, handed = false, hand_back = false;
                R top = R(0);
                R cc[4][4];  // position polynomial of this step (set when the step is accepted; read only on events)
                if (EEst2 != EEst2) {
                    done = RTGR_RAY_NAN;
                } else {
                    // ---- PI controller in log2 space: q = EEst^β1 / qold^β2 / γ ----------------------
                    const float le = 0.5f * flog2(fmax1(EEst2, 1e-37f));  // log2 of the error estimate itself (floor 3e-19)
                    float qf = fexp2(__builtin_fmaf(beta1, le, -beta2 * lq)) * igamma;
                    qf = (EEst2 == 0.0f) ? qmax_inv : fclamp1(qf, qmax_inv, qmin_inv);
        // [budget: reach bound]
                    bool hand_over = false;
                    bool need_scan = true;  // NEAR: false when no lane of the wave can see a sign change in this step
                    // NEAR: WHICH objects some lane of the wave may see change sign in this step — bit o >> mask_shift of the mask (one
                    // bit per object up to 64 objects; beyond, one per 2, 4, … neighbours of the device list, which the groups of a long
                    // list have made neighbours in space: a bit drags in what the ray is passing anyway), all ones for the passes that
                    // do not test reach.  The scan below folds only those objects into the sample points' minima: an
                    // object whose distance provably keeps its sign in this step, for every lane that takes part, cannot move the sign of
                    // the minimum (a ray outside every object stays outside the ones it cannot reach), and the sign is all the scan reads.
                    // With example2's three objects that saves little; with a list of 64 it is the difference between 9 x 64 distances
                    // per scanned step and 9 x the one or two spheres a ray is passing (2048², 64 objects, NEAR pass
                    // 65.9 -> see there).  Same results bit for bit: FULL == FAR + NEAR is under test with long lists too.
                    unsigned long long scan_mask = ~0ull;
                    // … and, for lists beyond 64 objects (where a bit stands for 2, 4, … 2048 neighbours), the positions themselves: up to
                    // RTGR_EARLY_BUF of them in the wave's LDS (the early list's buffer, idle in this pass); more than that — or "every
                    // object" — and the scan walks the mask's blocks instead
                    uint32_t ncand = RTGR_EARLY_BUF + 1u;
                    if constexpr (MODE == MODE_FAR || MODE == MODE_NEAR) {
                        if (EEst2 <= 1.0f) {
                            // ---- can ANY object's distance change sign anywhere in this step?  |x_q(θ) − x_q| <= δ_q for all
                            // θ in [0,1] (Nyström form of the dense output, beta[l] = max|B2_l(θ)|); a plane's distance then
                            // moves by <= δ_t, a sphere's by <= Σ_q δ_q (2|X_q| + δ_q), a disk's by <= max(δ_z, δ_x + δ_y).
                            R dl[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                R acc2 = N::beta[0] * rabs(k[0][q]);
#pragma unroll
                                for (int l = 1; l < 6; l++) acc2 = rfma(N::beta[l], rabs(KL(l, q)), acc2);
                                dl[q] = h * rfma(h, acc2, rabs(u[q]));
                            }
                            bool safe = true;
                            const R guard = R(1) + R(1e-6);
                            if constexpr (MODE == MODE_NEAR) {
                                // a lane INSIDE an object (ps < 0: the minimum is that object's negative distance and changes sign when
                                // the ray leaves it), or without a sign yet (ps == 0), needs every object: no selection for this wave-step
                                scan_mask = __ballot(run && !(ps > R(0))) != 0ull ? ~0ull : 0ull;
                                ncand = scan_mask != 0ull ? RTGR_EARLY_BUF + 1u : 0u;
                            }
                            // (+ an absolute floor of a few hundred ulp of the operands: a distance that is itself rounding noise must go
                            //  through the real scan)
                            auto note = [&](bool safe_o, uint32_t o) {
                                safe = safe && safe_o;
                                if constexpr (MODE == MODE_NEAR) {
                                    if (__ballot(run && !safe_o) != 0ull) {
                                        scan_mask |= 1ull << (o >> mask_shift);
                                        if (__builtin_expect(mask_shift != 0u, 0)) {
                                            if (ncand < RTGR_EARLY_BUF) early_buf[ncand] = o;   // (every lane that is here writes the same word)
                                            ncand++;
                                        }
                                    }
                                }
                            };
                            // (a lane that has ruled a whole GROUP of spheres out — see below — has ruled its members out: what the
                            //  members' own tests say for that lane, asked because another lane needs them, does not count)
                            bool group_out = false, run_out = false;
                            for_each_within_reach<R>(A.sc,
                              [&](const DevObject<R>& G, int level) -> bool {   // a group of a long list (level 0), or a run of groups (1): its bounding sphere
                                // OUTSIDE the bounding sphere now, and for the whole step (the sphere test below without the absolute
                                // value): then outside every member, by at least the same margin, for the whole step — no member's
                                // distance changes sign.  The floor keeps that margin above the rounding noise of the members' own
                                // distances (|X_g| − R_g > 64 eps (|X_g| + R_g), the noise is a few eps of |X_i| + R_i <= |X_g| + 2 R_g).
                                const R X0 = x[1] - G.p[1], X1 = x[2] - G.p[2], X2 = x[3] - G.p[3], Rr = G.p[8];
                                const R D0 = rfma(X0, X0, rfma(X1, X1, rfma(X2, X2, -Rr * Rr)));
                                const R B = rfma(dl[1], rfma(R(2), rabs(X0), dl[1]),
                                                 rfma(dl[2], rfma(R(2), rabs(X1), dl[2]), dl[3] * rfma(R(2), rabs(X2), dl[3])));
                                const R mag = rabs(D0) + R(2) * Rr * Rr;
                                const bool out = D0 > rfma(guard, B, R(256) * eps * mag);
                                if (level != 0) { run_out = out; group_out = out; }   // (a lane that has ruled the run out has ruled its groups out)
                                else group_out = run_out || out;
                                return __ballot(run && !group_out) != 0ull;
                              },
                              [&](const DevObject<R>& ob, uint32_t o) {   // the spheres of the list: no dispatch, one batch of scalar loads each
                                const R X0 = x[1] - ob.p[1], X1 = x[2] - ob.p[2], X2 = x[3] - ob.p[3], Rr = ob.p[8];
                                const R D0 = rfma(X0, X0, rfma(X1, X1, rfma(X2, X2, -Rr * Rr)));
                                const R B = rfma(dl[1], rfma(R(2), rabs(X0), dl[1]),
                                                 rfma(dl[2], rfma(R(2), rabs(X1), dl[2]), dl[3] * rfma(R(2), rabs(X2), dl[3])));
                                const R mag = rabs(D0) + R(2) * Rr * Rr;  // >= |X|² + R²: the operands' magnitude, for the floor
                                note(group_out || rabs(D0) > rfma(guard, B, R(256) * eps * mag), o);
                              },
                              [&](const DevObject<R>& ob, uint32_t o) {   // everything else
                                bool safe_o;
                                if (ob.kind == RTGR_PLANE) {
                                    safe_o = (rabs(x[0] - ob.p[0]) > rfma(guard, dl[0], R(256) * eps * (rabs(x[0]) + rabs(ob.p[0]))));
#ifdef RTGR_USER_OBJECTS
                                } else if (ob.kind == RTGR_USER_OBJECT) {
#ifdef RTGR_USER_REACH
                                    // the source's own bound of how far its distance can move inside the box |x'_q − x_q| <= δ_q
                                    // (+ a floor of a few hundred ulp of the distance's natural scale, 1 + |d|: a distance that is
                                    //  itself rounding noise goes through the real scan, as for the built-ins)
                                    const R gd[4] = {guard * dl[0], guard * dl[1], guard * dl[2], guard * dl[3]};
                                    const R du = rtgr_user_distance<R>(ob.type, x, ob.p);
                                    const R bu = rtgr_user_reach<R>(ob.type, x, ob.p, gd);
                                    safe_o = (rabs(du) > bu + R(256) * eps * (R(1) + rabs(du)));
#else
                                    safe_o = false;   // no bound given: never provably out of reach (such units run the FULL pass)
#endif
#endif
                                } else {   // RTGR_DISK
                                    R px = x[1], py = x[2];
                                    asm volatile("" : "+v"(px), "+v"(py));  // keep the disk's root inside this branch
                                    // (a maximum of three terms moves by at most the LARGEST of their moves: |z| by δ_z, the two radial
                                    //  terms by δ_ϱ <= δ_x + δ_y — not by their sum, which round 3 charged)
                                    safe_o = (rabs(disk_distance_fast<R>(ob, px, py, x[3])) > guard * rmax(dl[3], dl[1] + dl[2]));
                                }
                                note(safe_o, o);
                              });
                            if constexpr (MODE == MODE_FAR) hand_over = run && (!safe || (ps == R(0)));
                            else {
                                need_scan = !safe || (ps == R(0));
                                safe_streak = need_scan ? 0u : safe_streak + 1u;
                            }
                        }
                        // NEAR pass: a ray stays here until it ends, also after it has left every object's reach; its
                        // wave then runs with few lanes (the longest stays are 150-370 steps), so the scan is skipped
                        // whenever NO active lane needs it — a wave-uniform decision, same results by the same bound.
                        if constexpr (MODE == MODE_NEAR) {
                            need_scan = __ballot(run && (EEst2 <= 1.0f) && need_scan) != 0ull;
                            // (the mask was formed under the EXEC of the lanes that accepted their step, the same value in each of them:
                            //  read it from one of those — a lane that rejected still holds "every object", a superset and as correct —
                            //  and make it wave-uniform for the compiler too, so that the scan's per-object branches are scalar)
                            const unsigned long long m_acc = __ballot(EEst2 <= 1.0f);
                            scan_mask = uniform64(__shfl(scan_mask, m_acc != 0ull ? (int)__builtin_ctzll(m_acc) : 0, 64));
                            ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)ncand, m_acc != 0ull ? (int)__builtin_ctzll(m_acc) : 0, 64));
                        }
                    }
        // [budget: hand over]
                    if (hand_over) {
                        // hand the ray, in its PRE-step state, to the NEAR pass (which redoes this step with the full scan)
                        R* hd = A.hand + idx * HAND_W;
#pragma unroll
                        for (int q = 0; q < 4; q++) { hd[q] = x[q]; hd[4 + q] = u[q]; hd[8 + q] = k[0][q]; }
                        hd[12] = t; hd[13] = dt; hd[14] = ps; hd[15] = (R)lq;
                        const bool early_ray = MODE == MODE_FAR && A.early && nacc < A.near_early;
                        A.meta[idx * 3] = nacc; A.meta[idx * 3 + 1] = nrej;
                        A.meta[idx * 3 + 2] = early_ray ? META_HANDED_EARLY : META_HANDED;
                        // A ray handed over EARLY in its life is passing an object it may well miss, and then it stays in
                        // the NEAR pass for the rest of its path: in example2, 1.2 % of the rays, all handed over at
                        // steps 30-34, hold 55 % of the NEAR pass's steps.  They are listed so that the NEAR pass can
                        // start with them, packed together (collected per wave in LDS, see the end of the loop body).
                        if (early_ray) list_it = true;
                        handed = true;
        // [budget: controller]
                    } else if (EEst2 <= 1.0f) {
                        nacc++;
                        lq = fmax1(le, lq_init);  // log2(max(EEst, qoldinit))
                        const R dtnew = dt * (R)__builtin_amdgcn_rcpf(qf);
                        R tnew = t + dt;
                        // snap to λ1 within 10 ulp (OrdinaryDiffEq's fixed-tstop rule) — tested only where it can hold: λ ∈ [λ0, λ1], so
                        // |tnew − λ1| < 10 eps max(|tnew|, |λ1|) needs tnew > t1_snap (a wave-uniform constant): one compare per step
                        if (tnew > t1_snap) {
                            if (rabs(tnew - t1) < R(10) * eps * rmaxabs<R>(tnew, t1)) tnew = t1;
                        }
        // [budget: scan]
                        // ---- ContinuousCallback ----------------------------------------------------
                        // x(θ) = x + θ c1 + θ² c2 + θ³ c3 + θ⁴ c4 ;  c1 = h u,  c_m = h² Σ_l R2[l][m] k_l
                        // (FAR pass: proven above that no object's distance changes sign in this step — nothing to scan)
                        if (MODE != MODE_FAR && need_scan) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            cc[0][q] = h * u[q];
#pragma unroll
                            for (int m = 1; m < 4; m++)
                                cc[m][q] = h2 * rfma(N::R2[5][m], KL(5, q), rfma(N::R2[4][m], KL(4, q), rfma(N::R2[3][m], KL(3, q),
                                                rfma(N::R2[2][m], KL(2, q), rfma(N::R2[1][m], KL(1, q), N::R2[0][m] * k[0][q])))));
                        }
                        bool found = false;
                        R nextc;
                        // the objects of this step's scan: by the list of positions where there is one (NEAR, lists beyond 64 objects),
                        // else by the mask (for_each_masked_by_kind: bit tests for a list in the argument block, set bits beyond)
                        auto scan_objects = [&](auto&& fs, auto&& fo) {
                            if constexpr (MODE == MODE_NEAR) {
                                if (__builtin_expect(mask_shift != 0u && ncand <= RTGR_EARLY_BUF, 0)) {
                                    typedef const DevObject<R> __attribute__((address_space(4))) * ConstTable;
                                    const ConstTable table = (ConstTable)(unsigned long long)(A.sc.more - (uint32_t)RTGR_MAX_OBJECTS);
                                    for (uint32_t c_ = 0; c_ < ncand; c_++) {
                                        const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)early_buf[c_]);
                                        if (o < A.sc.nsph) fs(*(const DevObject<R>*)(table + o), o);
                                        else fo(*(const DevObject<R>*)(table + o), o);
                                    }
                                    return;
                                }
                            }
                            for_each_masked_by_kind<R>(A.sc, scan_mask, mask_shift, fs, fo);
                        };
                        if constexpr (NPTS10) {
                            // two blocks (θ = 1/9…5/9, then 6/9…8/9 + the end point) keep the live set inside the
                            // 256-register budget; each object's parameters are fetched once per block
                            {
                                R pos[5][4], dmin[5];
#pragma unroll
                                for (int j = 0; j < 5; j++) {
                                    const R th = R(j + 1) / R(9);
                                    dmin[j] = R(__builtin_huge_val());
#pragma unroll
                                    for (int q = 0; q < 4; q++)
                                        pos[j][q] = rfma(th, rfma(th, rfma(th, rfma(th, cc[3][q], cc[2][q]), cc[1][q]), cc[0][q]), x[q]);
                                }
                                scan_objects(
                                    [&](const DevObject<R>& ob, uint32_t) { fold_sphere<R, 5>(ob, pos, dmin); },
                                    [&](const DevObject<R>& ob, uint32_t) { fold_distances<R, 5>(ob, pos, dmin); });
#pragma unroll
                                for (int j = 0; j < 5; j++) {
                                    const bool hit = (ps * dmin[j] < R(0)) && !found;
                                    top = hit ? R(j + 1) / R(9) : top;
                                    found = found || hit;
                                }
                            }
                            {
                                R pos[4][4], dmin[4];
#pragma unroll
                                for (int j = 0; j < 3; j++) {
                                    const R th = R(j + 6) / R(9);
                                    dmin[j] = R(__builtin_huge_val());
#pragma unroll
                                    for (int q = 0; q < 4; q++)
                                        pos[j][q] = rfma(th, rfma(th, rfma(th, rfma(th, cc[3][q], cc[2][q]), cc[1][q]), cc[0][q]), x[q]);
                                }
                                dmin[3] = R(__builtin_huge_val());
#pragma unroll
                                for (int q = 0; q < 4; q++) pos[3][q] = xn[q];
                                scan_objects(
                                    [&](const DevObject<R>& ob, uint32_t) { fold_sphere<R, 4>(ob, pos, dmin); },
                                    [&](const DevObject<R>& ob, uint32_t) { fold_distances<R, 4>(ob, pos, dmin); });
                                nextc = dmin[3];
#pragma unroll
                                for (int j = 0; j < 3; j++) {
                                    const bool hit = (ps * dmin[j] < R(0)) && !found;
                                    top = hit ? R(j + 6) / R(9) : top;
                                    found = found || hit;
                                }
                            }
                        } else {
                            nextc = min_distance<R>(A.sc, xn);
                            const R dth = npts > 1 ? R(1) / R(npts - 1) : R(1);
                            for (int j = 1; j + 1 < npts; j++) {
                                const R th = R(j) * dth;
                                R xi[4];
#pragma unroll
                                for (int q = 0; q < 4; q++)
                                    xi[q] = rfma(th, rfma(th, rfma(th, rfma(th, cc[3][q], cc[2][q]), cc[1][q]), cc[0][q]), x[q]);
                                const bool hit = (ps * min_distance<R>(A.sc, xi) < R(0)) && !found;
                                top = hit ? th : top;
                                found = found || hit;
                            }
                        }
                        const bool endpoint = (ps != R(0)) && (ps * nextc <= R(0));
                        found = found && (ps != R(0)) && !endpoint;
                        if (endpoint || found) {
                            top = endpoint ? R(1) : top;
                            is_event = true;
                            is_interior = found;
                            done = RTGR_RAY_EVENT;
                        }
                        if (!is_event) ps = rsign(nextc);
                        }  // scan
        // [budget: controller]
                        if (!is_event) {
                            commit = true;
                            t = tnew;
                            dt = rmin(dtmax, dtnew);
                            if (!(t < t1)) done = RTGR_RAY_LAMBDA1;
                            else if (nacc + nrej >= A.opt.max_steps) done = RTGR_RAY_MAXSTEPS;
                            else if (!(t + dt > t)) done = RTGR_RAY_DTMIN;
                            else if (MODE == MODE_NEAR && A.allow_handback && safe_streak >= 2u && (nacc - nacc0) >= A.handback_after) hand_back = true;
                        }
                    } else {
                        nrej++;
                        // (EEst^β1, the rejected step's factor, is formed HERE: Float64 rays reject about one step in 10⁴, and an
                        //  exp2 + a product in every step's uniform flow were two issue slots — 3.6 with the transcendental's cost — for them)
                        dt = dt * (R)__builtin_amdgcn_rcpf(fmin1(qmin_inv, fexp2(beta1 * le) * igamma));
                        if (nacc + nrej >= A.opt.max_steps) done = RTGR_RAY_MAXSTEPS;
                        else if (!(t + dt > t)) done = RTGR_RAY_DTMIN;
                    }
                }
        // [budget: records and bookkeeping]
                if (!run) {  // a ghost lane decides nothing
                    done = 0xffu; is_event = false; is_interior = false; handed = false; hand_back = false; commit = false;
                    list_it = false;
                }
                // ---- an event: hand the step's position polynomial to the resolve kernel (built from the step-START state,
                // so it has to happen before the commit below) -----------------------------------------------------------------
                const bool want_state = (A.recw == REC_TAIL_STATE);
                if (is_event) {
                    const RecRef<R> rec{A.hand + idx * HAND_W, A.rec + idx * (uint64_t)A.recw};
#pragma unroll
                    for (int q = 0; q < 4; q++) rec[REC_X + q] = x[q];
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int q = 0; q < 4; q++) rec[REC_C + 4 * m + q] = cc[m][q];
                    rec[REC_PS] = ps;
                    rec[REC_TOP] = top;
                    rec[REC_T] = t;
                    rec[REC_H] = h;
                    if (want_state) {  // u(θ) = u + h Σ_j b_j(θ) k_j  (all seven stages)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            rec[REC_U + q] = u[q];
#pragma unroll
                            for (int m = 0; m < 4; m++)
                                rec[REC_CU + 4 * m + q] = h * rfma(N::r[6][m], k[6][q], rfma(N::r[5][m], KL(5, q),
                                    rfma(N::r[4][m], KL(4, q), rfma(N::r[3][m], KL(3, q), rfma(N::r[2][m], KL(2, q),
                                    rfma(N::r[1][m], KL(1, q), N::r[0][m] * k[0][q]))))));
                        }
                    }
                }
                // (the accepted step is committed AFTER this region, at the top level of the loop body)
                if (handed) state = L_FREE;
                if (hand_back) {
                    // NEAR -> next round's FAR pass: the ray has been out of every object's reach for two steps; its
                    // committed state goes back into the hand-over record and the cheaper pass carries it on
                    R* hd = A.hand + idx * HAND_W;
#pragma unroll
                    for (int q = 0; q < 4; q++) { hd[q] = xn[q]; hd[4 + q] = un[q]; hd[8 + q] = k[6][q]; }
                    hd[12] = t; hd[13] = dt; hd[14] = ps; hd[15] = (R)lq;
                    A.meta[idx * 3] = nacc; A.meta[idx * 3 + 1] = nrej; A.meta[idx * 3 + 2] = META_HANDBACK;
                    if (MODE == MODE_NEAR) c_maxnear = c_maxnear > (nacc - nacc0) ? c_maxnear : (nacc - nacc0);
                    state = L_FREE;
                }
                if (done != 0xffu) {
                    if (!is_event) {
                        // ended without an event (λ1, step cap, dt underflow, NaN): the state as it stands, θ = 0
                        const RecRef<R> rec{A.hand + idx * HAND_W, A.rec + idx * (uint64_t)A.recw};
#pragma unroll
                        for (int q = 0; q < 4; q++) rec[REC_X + q] = commit ? xn[q] : x[q];
                        rec[REC_PS] = R(0);
                        rec[REC_TOP] = R(0);
                        rec[REC_T] = t;
                        rec[REC_H] = R(0);
                        if (want_state) {
#pragma unroll
                            for (int q = 0; q < 4; q++) rec[REC_U + q] = commit ? un[q] : u[q];
                        }
                    }
                    uint32_t* mt = A.meta + idx * 3;
                    mt[0] = nacc;
                    mt[1] = nrej;
                    mt[2] = done | (is_interior ? 0x100u : 0u);
                    if (MODE == MODE_NEAR) c_maxnear = c_maxnear > (nacc - nacc0) ? c_maxnear : (nacc - nacc0);
#ifdef RTGR_ROOT_STATS
                    if (MODE == MODE_NEAR && A.dbg) {  // per ray: accepted steps at hand-over, accepted steps in this pass
                        uint32_t* pr = (uint32_t*)(A.dbg + 4ull * 8192) + 2 * idx;
                        pr[0] = nacc0; pr[1] = nacc - nacc0;
                    }
#endif
                    if constexpr (MODE == MODE_FAR) {
                        // a ray ENDING in the FAR pass is rare (λ1, step cap, NaN — never an event): count it with
                        // atomics of its own and keep the four accumulator registers out of the 3-waves/SIMD kernel
                        if (A.counters) {
                            atomicAdd(&A.counters[0], 1ull);
                            atomicAdd(&A.counters[1], (unsigned long long)nacc);
                            atomicAdd(&A.counters[2], (unsigned long long)nrej);
                            atomicAdd(&A.counters[3], 6ull * (nacc + nrej) + 2ull);
                            if (done >= RTGR_RAY_MAXSTEPS) atomicAdd(&A.counters[6], 1ull);
                        }
                    } else {
                        c_rays += 1; c_acc += nacc; c_rej += nrej;
                        c_ev += is_event; c_int += is_interior; c_nf += (done >= RTGR_RAY_MAXSTEPS);
                    }
                    state = L_FREE;
                }
            }
        }
        // [budget: commit]
        // ---- commit the accepted step: the only place (besides the refill) where the ray state is assigned ------------
        // (the compiler emits this as 12 v_mov_b64 under the lanes' EXEC mask — not as 24 v_cndmask_b32 selects; checked
        //  in the ISA, tools/isa_mix.py --list mov)
        if (commit) {
#pragma unroll
            for (int q = 0; q < 4; q++) { x[q] = xn[q]; u[q] = un[q]; k0[q] = k[6][q]; }
        }
        // [budget: early list]
        // ---- FAR: early-list entries are collected in LDS and appended to the global list RTGR_EARLY_BUF/2 at a time:
        // an append with its own device atomic stalls the wave for the atomic's round trip every time ANY lane has one
        // (measured: +13 % on the whole pass when a sixth of the rays qualified)
        if constexpr (MODE == MODE_FAR) {
            const unsigned long long m_list = __ballot(list_it);
            if (m_list != 0ull) {
                if (list_it) early_buf[e_cnt + mask_rank(m_list, lane)] = (uint32_t)idx;
                e_cnt += (uint32_t)__builtin_popcountll(m_list);
                if (e_cnt > RTGR_EARLY_BUF / 2) { flush_early(A, early_buf, e_cnt, lane); e_cnt = 0; }
            }
        }
        // [budget: end]
    }
    if constexpr (MODE == MODE_FAR) {
        if (e_cnt != 0u) flush_early(A, early_buf, e_cnt, lane);
    }
#ifdef RTGR_ROOT_STATS
    if (A.dbg && lane == 0) {
        unsigned long long* d = A.dbg + 4ull * blockIdx.x;
        d[0] = dbg_t0; d[1] = wall_clock64(); d[2] = dbg_iters; d[3] = dbg_rays;
    }
#endif
    if (MODE != MODE_FAR && A.counters) {
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
        if (MODE == MODE_NEAR) {  // diagnostics: the longest stay of any ray in the NEAR pass (accepted steps)
            unsigned long long mx = c_maxnear;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(mx, off, 64); mx = mx > o ? mx : o; }
            if (lane == 0) atomicMax(&A.counters[7], mx);
        }
    }
}

#undef KL
#undef KSTORE
template <class R, int METRIC, bool SPIN, bool NPTS10, int MODE>
__global__ __launch_bounds__(64, METRIC >= RTGR_GENERIC_BASE ? (sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD_GENERIC : RTGR_WAVES_PER_SIMD_GENERIC_F32)
                                 : (MODE == MODE_FAR ? (sizeof(R) == 8 ? (SPIN ? RTGR_WAVES_PER_SIMD_SPIN_FAR : RTGR_WAVES_PER_SIMD_FAR) : 4)
                                                     : (sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD : RTGR_WAVES_PER_SIMD_F32)))
void integrate_kernel(const IntegrateArgs<R> A) {
    integrate_body<R, METRIC, SPIN, NPTS10, MODE, (METRIC >= RTGR_GENERIC_BASE ? RTGR_LDSK_GENERIC != 0 : (SPIN && MODE == MODE_FAR && RTGR_LDSK_SPIN_FAR != 0))>(A);
}

// The a = 0 FAR pass once more at FOUR waves per SIMD (128 registers: 28 B/lane of scratch for KS_REF, none for KS_TRUE).
// Pays on big launches only — measured 84.5 vs 85.6 ms at 16.8 M rays, but 8.5 vs 7.3 ms at 1 M rays, where 4096
// persistent waves leave 4 rays per lane and the tail dominates; with spin (92 B/lane of scratch) it loses everywhere.
template <class R, int METRIC>
__global__ __launch_bounds__(64, 4) void integrate_far4_kernel(const IntegrateArgs<R> A) {
    integrate_body<R, METRIC, false, true, MODE_FAR>(A);
}

// Ray set-up, one thread per ray: the camera ray itself when the caller gave a camera instead of states (make_canvas,
// :457-478 — the state never goes through HBM), its key for the longest-first queue (see "ray ordering" below), u̇(y0), the Hairer initial step (d0, d1, one Euler probe, d2; the
// norms in f32 like the controller's), sign(min_distance(y0)) for the ContinuousCallback (App. B.4) and the controller's
// q_old = 1e-4 (App. B.2) -> the ray's 16-scalar start record.  2 RHS evaluations per ray (counted by the integrate
// kernel's counters).  A body function for the same reason as integrate_body.
template <class R>
RTGR_DEV void order_key(const R x4[4], const R u4[4], bool valid, uint64_t w, uint8_t* keys, uint32_t* hist);

template <class R, int METRIC, bool SPIN>
RTGR_DEV void prepare_body(const IntegrateArgs<R>& A) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = w < A.n;
    const MetricK<R> MK = metric_consts<R>(A.sc.M, A.sc.a);
    const R reltol = A.opt.reltol, abstol = A.opt.abstol, dtmax = A.opt.lambda1 - A.opt.lambda0;
    R x[4] = {R(0), R(1), R(0), R(0)}, u[4] = {R(-1), R(0), R(1), R(0)}, k1[4], k2[4];
    if (valid) {
        if (A.state0) {
            const R* s0 = A.state0 + w * 8;
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 4; q++) { x[q] = s0[q]; u[q] = s0[4 + q]; bad = bad || x[q] != x[q] || u[q] != u[q]; }
            // `@assert !any(isnan, xx)` of kerr_schild (src/RayTraceGR.jl:279): evaluated here, where every caller-supplied
            // state is read anyway; the host entry points turn the flag into RTGR_ERR_NAN_INPUT
            if (bad && A.nan_flag) atomicOr(A.nan_flag, 1u);
        } else {  // make_canvas (src/RayTraceGR.jl:457-478) for this pixel, straight into registers
            R s[8];
            const uint64_t idx = A.first + w;
            make_pixel<R>(A.sc, A.cam, A.ni, A.nj, idx % A.ni, A.j0 + (idx / A.ni) * A.jstride, s);
#pragma unroll
            for (int q = 0; q < 4; q++) { x[q] = s[q]; u[q] = s[4 + q]; }
        }
    }
    if (A.keys) order_key<R>(x, u, valid, w, A.keys, A.hist);  // block-wide (LDS histogram): before any early exit
    if (!valid) return;
    // The ray's state is a VALUE from here on, whichever way it was obtained: without this barrier the compiler contracts
    // the last product of make_pixel (u = (…)·1/√2) into the first sum of the RHS (k_a u^a = u^t + …) when the camera ray
    // is generated in this kernel, and cannot when the same ray is loaded from the caller's array — camera and state0
    // frames then differ in the last bit of u̇(y0) (found by test_host_pipeline_with_many_chunks_and_every_output).
#pragma unroll
    for (int q = 0; q < 4; q++) { asm volatile("" : "+v"(x[q])); asm volatile("" : "+v"(u[q])); }
    accel<R, METRIC, SPIN, true>(x + 1, u, MK, k1, x[0]);      // f0 = (u, k1)
    float acc0 = 0.0f, acc1 = 0.0f;
    float iskx[4], isku[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        iskx[q] = __builtin_amdgcn_rcpf((float)rfma(rabs(x[q]), reltol, abstol));
        isku[q] = __builtin_amdgcn_rcpf((float)rfma(rabs(u[q]), reltol, abstol));
        const float a0 = (float)x[q] * iskx[q], b0 = (float)u[q] * isku[q];
        const float a1 = (float)u[q] * iskx[q], b1 = (float)k1[q] * isku[q];
        acc0 = __builtin_fmaf(a0, a0, __builtin_fmaf(b0, b0, acc0));
        acc1 = __builtin_fmaf(a1, a1, __builtin_fmaf(b1, b1, acc1));
    }
    const float d0f = __builtin_sqrtf(acc0 * 0.125f), d1f = __builtin_sqrtf(acc1 * 0.125f);
    const float dt0f = (d0f < 1e-5f || d1f < 1e-5f) ? 1e-6f : (d0f / d1f) * 0.01f;
    const R dt0 = rmin((R)dt0f, dtmax);
    R X[3], U[4];                                              // y0 + dt0 f0
#pragma unroll
    for (int q = 0; q < 4; q++) U[q] = rfma(dt0, k1[q], u[q]);
#pragma unroll
    for (int q = 0; q < 3; q++) X[q] = rfma(dt0, u[1 + q], x[1 + q]);
    accel<R, METRIC, SPIN, true>(X, U, MK, k2, rfma(dt0, u[0], x[0]));          // f1 − f0 = (dt0·k1, k2 − k1)
    float acc2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float a2 = (float)(dt0 * k1[q]) * iskx[q], b2 = (float)(k2[q] - k1[q]) * isku[q];
        acc2 = __builtin_fmaf(a2, a2, __builtin_fmaf(b2, b2, acc2));
    }
    const float d2f = __builtin_sqrtf(acc2 * 0.125f) / (float)dt0;
    const float md = fmaxf(d1f, d2f);
    // dt1 = 10^(-(2 + log10 md)/5) = 2^(-(2 log2 10 + log2 md)/5)
    const float dt1f = (md <= 1e-15f) ? fmaxf(1e-6f, (float)dt0 * 1e-3f) : fexp2(-0.2f * (6.643856189774724f + flog2(md)));
    const R dt_init = rmin(rmin(R(100) * dt0, (R)dt1f), dtmax);
    R* hd = A.hand + w * HAND_W;
#pragma unroll
    for (int q = 0; q < 4; q++) { hd[q] = x[q]; hd[4 + q] = u[q]; hd[8 + q] = k1[q]; }
    hd[12] = A.opt.lambda0;
    hd[13] = dt_init;
    hd[14] = rsign(min_distance<R>(A.sc, x));
    hd[15] = R(-13.287712379549449);                           // log2(qoldinit = 1e-4)
}

template <class R, int METRIC, bool SPIN>
__global__ __launch_bounds__(256) void prepare_kernel(const IntegrateArgs<R> A) {
    prepare_body<R, METRIC, SPIN>(A);
}

// Per-launch reset of the queue heads and of the ordering histogram.  A kernel rather than hipMemsetAsync: memset nodes
// of a captured HIP graph were observed not to re-run on later replays (ROCm 7.0 runtime bundled with PyTorch), which
// left stale queue heads / histograms and sent the scatter out of bounds; kernel nodes replay reliably.
static __global__ __launch_bounds__(256) void reset_kernel(unsigned long long* ctrl, uint32_t* hist512) {
    if (threadIdx.x < 8) ctrl[threadIdx.x] = 0ull;
    if (hist512) { hist512[threadIdx.x] = 0u; hist512[256 + threadIdx.x] = 0u; }
}

// ---------------------------------------------------------------------------------------------------------------------
// ray ordering: longest-expected-first (LPT) queue order
//
// Rays need 27…991 step attempts and a lane processes its rays one after another, so with few rays per lane (small
// screens, or one slab of an 8-GPU split: ~10 rays per lane) the kernel time is set by the lanes that happen to draw a
// long ray LAST: greedy scheduling in natural order runs 1.24x (10 rays/lane) … 1.46x (5 rays/lane) over the ideal.
// The rays that get long are the ones aimed at the hole, so the queue is ordered by the angle α between the ray and the
// direction to the origin (sin α = impact parameter / distance): a 256-bucket counting sort, ascending.  Simulated
// makespan over ideal with that order: 1.02-1.03.  The order only changes WHEN a ray is integrated, never its result.
// ---------------------------------------------------------------------------------------------------------------------
template <class R>
RTGR_DEV void order_key(const R x4[4], const R u4[4], bool valid, uint64_t w, uint8_t* keys, uint32_t* hist) {
    __shared__ uint32_t lh[256];  // called by every thread of a 256-thread block (prepare_kernel)
    lh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t b = 0xffffffffu;
    if (valid) {
        const R x = x4[1], y = x4[2], z = x4[3], ux = u4[1], uy = u4[2], uz = u4[3];
        const R xx = x * x + y * y + z * z, uu = ux * ux + uy * uy + uz * uz, xu = x * ux + y * uy + z * uz;
        float sin2 = 1.0f;
        if (xu < R(0) && xx > R(0) && uu > R(0)) sin2 = fmaxf(0.0f, 1.0f - (float)(xu * xu / (xx * uu)));
        b = (uint32_t)fminf(255.0f, 256.0f * __builtin_sqrtf(sin2));  // moving away -> last bucket
        keys[w] = (uint8_t)b;
    }
    // neighbouring rays share a handful of buckets: one LDS atomic per distinct bucket per wave, not one per ray
    unsigned long long todo = __ballot(valid);
    while (todo != 0ull) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t b0 = __shfl(b, (int)leader, 64);
        const unsigned long long m = __ballot(valid && b == b0);
        if ((threadIdx.x & 63) == leader) atomicAdd(&lh[b0], (uint32_t)__builtin_popcountll(m));
        todo &= ~m;
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
// exclusive prefix sum of the 256-bin histogram (one block) -> running offsets used by the scatter
static __global__ __launch_bounds__(256) void order_scan_kernel(const uint32_t* hist, uint32_t* offsets) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int b = 0; b < 256; b++) { const uint32_t c = sh[b]; sh[b] = acc; acc += c; }
    }
    __syncthreads();
    offsets[threadIdx.x] = sh[threadIdx.x];
}
// order[offset(bucket)++] = ray index.  Ranks inside a 256-ray block come from LDS atomics; each block then claims its
// range of every non-empty bucket with ONE global atomic (neighbouring rays share a handful of buckets).
static __global__ __launch_bounds__(256) void order_scatter_kernel(const uint8_t* keys, uint64_t n, uint32_t* offsets, uint32_t* order) {
    __shared__ uint32_t lcount[256], gbase[256];
    lcount[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = w < n;
    const uint32_t b = valid ? keys[w] : 0xffffffffu;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t r = 0;
    unsigned long long todo = __ballot(valid);
    while (todo != 0ull) {  // one LDS atomic per distinct bucket per wave; lanes rank themselves inside the ballot mask
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t b0 = __shfl(b, (int)leader, 64);
        const unsigned long long m = __ballot(valid && b == b0);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&lcount[b0], (uint32_t)__builtin_popcountll(m));
        base = __shfl(base, (int)leader, 64);
        if (valid && b == b0) r = base + mask_rank(m, lane);
        todo &= ~m;
    }
    __syncthreads();
    if (lcount[threadIdx.x]) gbase[threadIdx.x] = atomicAdd(&offsets[threadIdx.x], lcount[threadIdx.x]);
    __syncthreads();
    if (valid) order[gbase[b] + r] = (uint32_t)w;
}

// ---------------------------------------------------------------------------------------------------------------------
// resolve kernel: one thread per ray — root of cond(x(θ)) on [0, top], end state, colouring rule, stores
// ---------------------------------------------------------------------------------------------------------------------
template <class R, bool SEL = false>
RTGR_DEV R cond_poly(const DevScene<R>& sc, const R x0[4], const R c[4][4], R th, ObjSel sel = ObjSel{}) {
    R x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = rfma(th, rfma(th, rfma(th, rfma(th, c[3][q], c[2][q]), c[1][q]), c[0][q]), x0[q]);
    return min_distance<R, SEL>(sc, x, sel);
}

// LONG LISTS.  The root-find below evaluates the condition — the minimum over ALL objects' distances (:433-441) — some ten times per
// event, and the colour rule once more: with N objects, ~10 N distances per ray, of which all but one or two are of objects nowhere
// near the step the event lies in.  For a list beyond the argument block the wave first narrows the list down: over the event's step
// every object's distance stays within d_i(0) ± B_i (the bounds of the FAR pass's reach test, from the box |x_q(θ) − x_q(0)| <= δ_q,
// θ in [0, top], that the step's polynomial spans), so the minimum never exceeds U = min_i (d_i(0) + B_i), and an object with
// d_i(0) − B_i > U is never the minimum — neither its value nor its index can enter a result, at any θ of the bracket.  The objects
// that SOME lane of the wave cannot leave out form the selection (one bit per object up to 64, per 2^shift neighbours beyond); the
// rays of a wave are neighbours on the canvas, so the selection is a handful of objects.  Same frame bit for bit (under test with
// option groups = 0, which switches this off too).  Guards as in the reach test: 1e-6 relative on the bound, a floor of 256 ulp of the
// operands' magnitude; a NaN anywhere keeps the object in.
template <class R>
RTGR_DEV void distance_bounds(const DevObject<R>& ob, const R x[4], const R dl[4], R* lower, R* upper) {
    const R eps = sizeof(R) == 8 ? R(2.220446049250313e-16) : R(1.1920929e-7);
    const R guard = R(1) + R(1e-6);
    R d0, B, mag;
    if (ob.kind == RTGR_SPHERE) {
        const R X0 = x[1] - ob.p[1], X1 = x[2] - ob.p[2], X2 = x[3] - ob.p[3], Rr = ob.p[8];
        const R D0 = rfma(X0, X0, rfma(X1, X1, rfma(X2, X2, -Rr * Rr)));
        B = rfma(dl[1], rfma(R(2), rabs(X0), dl[1]), rfma(dl[2], rfma(R(2), rabs(X1), dl[2]), dl[3] * rfma(R(2), rabs(X2), dl[3])));
        mag = rabs(D0) + R(2) * Rr * Rr;
        d0 = Rr < R(0) ? -D0 : D0;
    } else if (ob.kind == RTGR_PLANE) {
        d0 = x[0] - ob.p[0];
        B = dl[0];
        mag = rabs(x[0]) + rabs(ob.p[0]);
#ifdef RTGR_USER_OBJECTS
    } else if (ob.kind == RTGR_USER_OBJECT) {
#ifdef RTGR_USER_REACH
        const R gd[4] = {guard * dl[0], guard * dl[1], guard * dl[2], guard * dl[3]};
        d0 = rtgr_user_distance<R>(ob.type, x, ob.p);
        B = rtgr_user_reach<R>(ob.type, x, ob.p, gd);
        mag = R(1) + rabs(d0);
#else
        d0 = R(0); B = R(__builtin_huge_val()); mag = R(0);   // no bound given: always in, never bounds the minimum
#endif
#endif
    } else {   // RTGR_DISK: a maximum of three terms moves by at most the largest of their moves
        d0 = obj_distance<R>(ob, x);
        B = rmax(dl[3], dl[1] + dl[2]);
        mag = rabs(d0) + rabs(x[1]) + rabs(x[2]) + rabs(x[3]) + rabs(ob.p[2]);
    }
    const R w = rfma(guard, B, R(256) * eps * mag);
    *lower = d0 - w;
    *upper = d0 + w;
}
template <class R>
RTGR_DEV ObjSel select_objects(const DevScene<R>& sc, const R x0[4], const R c[4][4], R top, bool event) {
    ObjSel sel{0ull, objsel_shift(sc.nobj), 0u, 0u};
    R dl[4];
#pragma unroll
    for (int q = 0; q < 4; q++) dl[q] = top * rfma(top, rfma(top, rfma(top, rabs(c[3][q]), rabs(c[2][q])), rabs(c[1][q])), rabs(c[0][q]));
    R U = R(__builtin_huge_val());
    auto upper = [&](const DevObject<R>& ob, uint32_t) {
        R lo, up;
        distance_bounds<R>(ob, x0, dl, &lo, &up);
        U = up < U ? up : U;
    };
    if (sc.ngroups == 0u) {   // a list without groups: every object twice
        for_each_object<R>(sc, upper);
        for_each_object<R>(sc, [&](const DevObject<R>& ob, uint32_t o) {
            R lo, up;
            distance_bounds<R>(ob, x0, dl, &lo, &up);
            if (__ballot(event && !(lo > U)) != 0ull) sel.add(o);
        });
        return sel;
    }
    // A grouped list (DevScene): the bound U may come from ANY objects — from a sample first (one member of every group, the loose
    // spheres, the other kinds).  A group whose bounding sphere stays farther than sqrt(U) away for the whole step holds no member
    // that can be the minimum: with m = |X_g| − R_g − |δ| > sqrt(U) >= 0 every member has |X_i(θ)| − r_i >= m, so its distance
    // (|X_i| − r_i)(|X_i| + r_i) >= m² > U (members have r_i >= 0: the host leaves inside-out spheres loose).  Such groups — and
    // runs of groups — are passed over when every lane agrees; the members that remain tighten the bound (U2: the minimum over the
    // sample AND over everything that passed, which holds the object the full list's bound comes from), and a last walk over what
    // passed keeps what the tight bound cannot leave out: the same selection as two walks over the whole list give.
    for_each_sample<R>(sc, upper);
    const R eps = sizeof(R) == 8 ? R(2.220446049250313e-16) : R(1.1920929e-7);
    const R guard = R(1) + R(1e-6);
    const R reach = guard * (rsqrt_(rfma(dl[1], dl[1], rfma(dl[2], dl[2], dl[3] * dl[3]))) + (U > R(0) ? rsqrt_(U) : R(0)));
    R U2 = U;
    auto member = [&](const DevObject<R>& ob, uint32_t o) {
        R lo, up;
        distance_bounds<R>(ob, x0, dl, &lo, &up);
        const bool in = !(lo > U);
        U2 = (in && up < U2) ? up : U2;
        if (__ballot(event && in) != 0ull) sel.add(o);
    };
    for_each_within_reach<R>(sc,
        [&](const DevObject<R>& G, int) -> bool {
            const R X0 = x0[1] - G.p[1], X1 = x0[2] - G.p[2], X2 = x0[3] - G.p[3];
            const R S = rfma(X0, X0, rfma(X1, X1, X2 * X2));
            const R t = G.p[8] + reach;
            const R rhs = t * t;
            const bool far_away = S > rfma(guard, rhs, R(256) * eps * (S + rhs));
            return __ballot(event && !far_away) != 0ull;
        },
        member, member);
    ObjSel fin{0ull, sel.shift, 0u, 0u};
    for_each_selected<R>(sc, sel, [&](const DevObject<R>& ob, uint32_t o) {
        R lo, up;
        distance_bounds<R>(ob, x0, dl, &lo, &up);
        if (__ballot(event && !(lo > U2)) != 0ull) fin.add(o);
    });
    return fin;
}

// Bracketed root of g(θ) = ps·cond(x(θ)) on [0, top], g(0) > 0 >= g(top).  Ridders' method: every iterate stays inside
// the bracket, the estimate x4 converges quadratically (the bracket WIDTH only halves per iteration, so convergence is
// judged on successive estimates).  Once the estimate has settled, probes 16 ulp before and after it pin the crossing:
// the result is a point with g >= 0 within ~32 ulp of it — the reference's prevfloat(find_zero(...))
// up to a few ulp (a 512-ulp window, 1e-13 in θ, for the rays whose distance is too noisy for that).  If the probes fail
// (estimate was off) the loop simply continues on the tightened bracket; the bisection point `mid` guarantees progress.
// -DRTGR_ROOT_STATS builds report the iteration count of every ray through lambda_end.
template <class R, bool SEL = false>
RTGR_DEV R event_root(const DevScene<R>& sc, const R x0[4], const R c[4][4], R ps, R top, int* iters = nullptr, ObjSel sel = ObjSel{}) {
    R lo = R(0), hi = top;
    R fhi = cond_poly<R, SEL>(sc, x0, c, hi, sel) * ps;
    R flo = cond_poly<R, SEL>(sc, x0, c, R(0), sel) * ps;
    R result = R(0);
    bool done = false;
    if (fhi == R(0)) { result = hi; done = true; }
    else if (!(flo > R(0)) || !(fhi < R(0))) { result = R(0); done = true; }
    const R eps = sizeof(R) == 8 ? R(2.220446049250313e-16) : R(1.1920929e-7);
    R est_prev = R(-1);
    for (int it = 0; it < 96 && !done; it++) {
        if (iters) *iters = it + 1;
        const R width = hi - lo;
        const R mid = rfma(R(0.5), width, lo);
        if (!(width > R(2) * eps * hi) || !(mid > lo && mid < hi)) {
            result = lo; done = true;
        } else {
            const R fm = cond_poly<R, SEL>(sc, x0, c, mid, sel) * ps;
            const R rad = rfma(fm, fm, -flo * fhi);  // > 0 since flo > 0 > fhi
            // Ridders' estimate.  When one end of the bracket already sits on the root (|g(lo)| ~ 1e-17 after a lucky
            // iterate) the formula returns that end itself: the point to EVALUATE is then the midpoint (progress), but the
            // ESTIMATE is the end — without this distinction such rays never "settled" and bisected 45 more times
            // (0.3 % of the rays, 13-50 iterations; their waves waited: 4 iterations per ray, 10.5 per wave).
            const R xr = rfma((mid - lo) * fm, frsq<R>(rad), mid);
            const bool inside = xr > lo && xr < hi;
            const R x4 = inside ? xr : mid;
            const R est = inside ? xr : (xr <= lo ? lo : (xr >= hi ? hi : mid));
            const R f4 = (x4 == mid) ? fm : cond_poly<R, SEL>(sc, x0, c, x4, sel) * ps;
            // An exact zero is common (a plane at a representable time makes g vanish on a whole ulp-interval of θ):
            // "directly at zero" is an accepted result, so stop there.
            if (fm == R(0)) { result = mid; done = true; }
            else if (f4 == R(0)) { result = x4; done = true; }
            const R a = rmin(mid, x4), b = rmax(mid, x4);
            const R fa = (mid <= x4) ? fm : f4, fb = (mid <= x4) ? f4 : fm;
            if (fa > R(0)) {
                lo = a; flo = fa;
                if (fb > R(0)) { lo = b; flo = fb; } else { hi = b; fhi = fb; }
            } else {
                hi = a; fhi = fa;
            }
            // two successive estimates within 256 ulp: converged down to the rounding noise of the distance itself
            // (ulp of max(θ, top/16): the noise is absolute in θ — λ = t + hθ is what matters — so a root near θ = 0
            //  must not be chased to ITS ulp)
            const R scale = rmax(est, R(0.0625) * top);
            const bool settled = RTGR_ROOT_SHORTCUT && (rabs(est - est_prev) <= R(256) * eps * scale);
            est_prev = est;
            if (settled && !done) {
                // Verify the settled estimate two-sidedly: 16 ulp before and after; if the sign change is not in there
                // (the distance is evaluated with a rounding noise of ~10 ulp of θ, which makes the estimates jitter and
                // can push the crossing out), 512 ulp.  Rays that never pass fall back to ~50 bisection steps on the
                // one-sided Ridders bracket, and their whole wave waits for them (measured with an 8-ulp settle test
                // and the 16-ulp window only: mean 5.4 iterations per ray, 15.7 per wave).
#pragma unroll 1
                for (int pass = 0; pass < 2 && !done; pass++) {
                    const R wd = (pass == 0 ? R(16) : R(512)) * eps;
                    const R pm = rmax(rfma(-wd, scale, est), lo), pp = rmin(rfma(wd, scale, est), hi);
                    const R fpm = (pm > lo) ? cond_poly<R, SEL>(sc, x0, c, pm, sel) * ps : flo;
                    const R fpp = (pp < hi) ? cond_poly<R, SEL>(sc, x0, c, pp, sel) * ps : fhi;
                    if (!(fpm < R(0)) && !(fpp > R(0))) {
                        result = pm; done = true;  // the sign change (or an exact zero at pm) is inside [pm, pp]
                    } else if (fpp == R(0)) {
                        result = pp; done = true;
                    } else {
                        // the crossing is elsewhere: tighten the bracket with what was learnt
                        if (fpm > R(0)) { lo = pm; flo = fpm; } else { hi = pm; fhi = fpm; }
                        if (fpp > R(0)) { if (pp > lo) { lo = pp; flo = fpp; } } else if (pp < hi) { hi = pp; fhi = fpp; }
                    }
                }
                if (!done) est_prev = R(-1);
            }
        }
    }
    return done ? result : lo;
}
#ifdef RTGR_ROOT_STATS
#define RTGR_ROOT_STATS_ARG , &root_iters
#else
#define RTGR_ROOT_STATS_ARG , nullptr
#endif

// the stores of one resolved ray
template <class R>
RTGR_DEV void resolve_store(const ResolveArgs<R>& A, uint64_t w, const RecRef<const R>& rec, const R xe[4], R Theta, R t, R h, const R col[3],
                            uint32_t hit, R ps, int root_iters) {
    (void)root_iters;
    const uint64_t idx = A.offset + w;
    A.rgb[idx] = col[0];
    A.rgb[A.n_slab + idx] = col[1];
    A.rgb[2 * A.n_slab + idx] = col[2];
    const uint32_t* mt = A.meta + w * 3;
    if (A.state_end) {
        R* se = A.state_end + idx * 8;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            se[q] = xe[q];
            R ue = rec[REC_U + q];
            if (ps != R(0)) {
                const R* cu = rec.tail + (REC_CU - HAND_W);
                ue = rfma(Theta, rfma(Theta, rfma(Theta, rfma(Theta, cu[12 + q], cu[8 + q]), cu[4 + q]), cu[q]), ue);
            }
            se[4 + q] = ue;
        }
    }
#ifdef RTGR_ROOT_STATS
    if (A.lambda_end) A.lambda_end[idx] = (R)root_iters;  // debug build: iterations of the root find
#else
    if (A.lambda_end) A.lambda_end[idx] = rfma(h, Theta, t);
#endif
    if (A.status) A.status[idx] = (uint8_t)(mt[2] & 0xffu);
    if (A.hit) A.hit[idx] = (uint8_t)hit;
    if (A.hit32) A.hit32[idx] = hit;
    if (A.n_accept) A.n_accept[idx] = mt[0];
    if (A.n_reject) A.n_reject[idx] = mt[1];
}
template <class R> RTGR_DEV void resolve_body_selected(const ResolveArgs<R>& A);

// (a body function: a unit with user objects wraps it in a resolve kernel of its own — the root-find evaluates the objects'
//  distances and the colour rule their objcolor)
template <class R>
RTGR_DEV void resolve_body(const ResolveArgs<R>& A) {
    if (__builtin_expect(A.select != 0u && A.sc.nobj > (uint32_t)RTGR_MAX_OBJECTS && A.n != 0, 0)) { resolve_body_selected<R>(A); return; }
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= A.n) return;
    const RecRef<const R> rec{A.hand + w * HAND_W, A.rec + w * (uint64_t)A.recw};
    R x0[4], xe[4];
#pragma unroll
    for (int q = 0; q < 4; q++) xe[q] = x0[q] = rec[REC_X + q];
    const R ps = rec[REC_PS], top = rec[REC_TOP], t = rec[REC_T], h = rec[REC_H];
    R Theta = R(0);
    int root_iters = 0;
    (void)root_iters;
    if (ps != R(0)) {  // an event: the polynomial part of the record is valid
        R c[4][4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int q = 0; q < 4; q++) c[m][q] = rec[REC_C + 4 * m + q];
        Theta = event_root<R>(A.sc, x0, c, ps, top RTGR_ROOT_STATS_ARG);
#pragma unroll
        for (int q = 0; q < 4; q++)
            xe[q] = rfma(Theta, rfma(Theta, rfma(Theta, rfma(Theta, c[3][q], c[2][q]), c[1][q]), c[0][q]), x0[q]);
    }
    R col[3];
    const uint32_t hit = colour_pixel<R>(A.sc, A.opt, xe, col);
    resolve_store<R>(A, w, rec, xe, Theta, t, h, col, hit, ps, root_iters);
}
// … and the same with the list narrowed down first (select_objects): lists beyond the argument block
template <class R>
RTGR_DEV void resolve_body_selected(const ResolveArgs<R>& A) {
    const uint64_t w0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = w0 < A.n;
    const uint64_t w = live ? w0 : A.n - 1;   // (every lane of the wave takes part in the selection's ballots; the spare ones repeat the last ray, silently)
    const RecRef<const R> rec{A.hand + w * HAND_W, A.rec + w * (uint64_t)A.recw};
    R x0[4], xe[4];
#pragma unroll
    for (int q = 0; q < 4; q++) xe[q] = x0[q] = rec[REC_X + q];
    const R ps = rec[REC_PS], top = rec[REC_TOP], t = rec[REC_T], h = rec[REC_H];
    R Theta = R(0);
    int root_iters = 0;
    (void)root_iters;
    const bool event = ps != R(0);
    R c[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) c[m][q] = event ? rec[REC_C + 4 * m + q] : R(0);
    ObjSel sel = select_objects<R>(A.sc, x0, c, event ? top : R(0), event);
    if (event) {
        Theta = event_root<R, true>(A.sc, x0, c, ps, top RTGR_ROOT_STATS_ARG, sel);
#pragma unroll
        for (int q = 0; q < 4; q++)
            xe[q] = rfma(Theta, rfma(Theta, rfma(Theta, rfma(Theta, c[3][q], c[2][q]), c[1][q]), c[0][q]), x0[q]);
    }
    // (a ray without an event is coloured where it stopped: nothing is known about that point — every object is asked)
    if (__ballot(!event) != 0ull) { sel.mask = ~0ull; sel.count = 65u; }
    R col[3];
    const uint32_t hit = colour_pixel<R, true>(A.sc, A.opt, xe, col, sel);
    if (live) resolve_store<R>(A, w, rec, xe, Theta, t, h, col, hit, ps, root_iters);
}
template <class R>
__global__ __launch_bounds__(256) void resolve_kernel(const ResolveArgs<R> A) {
    resolve_body<R>(A);
}

}  // namespace rtgr
