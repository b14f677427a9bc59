// rtgr_pipeline.hpp — host side of the trace pipeline: launch policy and the per-chunk kernel sequence
//     reset -> prepare (camera ray, queue key, u̇(y0), initial dt) -> [order scan/scatter] -> integrate<FAR> -> integrate<NEAR>
//     -> resolve
// as templates over (scalar, metric variant, spin).  Each tu_*.hip instantiates the variants of its group, so the
// library's translation units build in parallel and the host units (rtgr_internal.hpp lists them) instantiate no kernel at all.
//
// Launch policy (Knobs, rtgr_host.hpp; "auto" decisions by launch size are documented where they are taken):
//   waves_per_cu       resident waves per CU of the integrate kernel (auto = 4 x the instantiation's waves/SIMD)
//   chunk              rays per pipeline chunk (auto 2^26, less if memory is short); bounds the workspace
//   split = 0          one FULL integrate pass instead of the FAR + NEAR pair
//   order = 0 / 1      natural ray order / longest-expected-first (rtgr_persistent.hpp); auto: ordered, except Float32 above 2 M rays
//   fair = s           time slice 2^s clocks of the priority rotation (0 = off; auto 13 for 0.8-1.8 M rays, else off)
//   near_early = n     accepted steps at hand-over below which a ray is put on the NEAR pass's early list (64)
//   waves_per_cu_near  resident waves per CU of the NEAR pass (auto 4 below 2.4 M rays — 6.3 M with spin —, else all)
//   far4 = 0/1         force the 3- / 4-waves-per-SIMD instantiation of the a = 0 FAR pass (auto: by launch size)
//   pack = 0           Float32: the scalar one-ray-per-lane kernel instead of the packed two-rays-per-lane one
//   packfar = 1        Float32 experiment: packed scan-free FAR pass (3 waves/SIMD) + scalar NEAR pass instead of the one FULL pass
#pragma once
#include "rtgr_host.hpp"
#include "rtgr_persistent.hpp"
#include "rtgr_packed_f32.hpp"

namespace rtgr {

#ifndef RTGR_F32_SPLIT_FROM
#define RTGR_F32_SPLIT_FROM 32
#endif

// One lane = one ray.  A wave owns an 8x8 pixel tile (lock-step efficiency 0.90 vs 0.45 for 64 consecutive
// pixels); a 256-thread workgroup owns 4 horizontally adjacent tiles.  The simple variant (knob tile = 1):
// whole adaptive loop + event finder + colouring inline — an independent formulation kept for A/B and cross-checks.
template <class R, int METRIC, bool SPIN>
__global__ __launch_bounds__(256) void trace_kernel(const TraceArgs<R> A) {
    const uint64_t tiles_i = (A.ni + 7) >> 3;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t ti = wave % tiles_i, tj = wave / tiles_i;
    const uint64_t i = ti * 8 + (lane & 7), jl = tj * 8 + (lane >> 3);
    const bool valid = (i < A.ni) && (jl < A.nrows);
    const uint64_t n = A.ni * A.nrows;
    const uint64_t idx = i + jl * A.ni;

    RayStats st{0, 0, 0, 0, 0};
    bool ev = false;
    if (valid) {
        R s0[8], se[8], lam, col[3];
        if (A.state0) {
#pragma unroll
            for (int c = 0; c < 8; c++) s0[c] = A.state0[idx * 8 + c];
        } else {
            make_pixel<R>(A.sc, A.cam, A.ni, A.nj, i, A.j0 + jl * A.jstride, s0);
        }
        st = integrate_ray<R, METRIC, SPIN>(A.sc, A.opt, s0, se, lam);
        const uint32_t hit = colour_pixel<R>(A.sc, A.opt, se, col);
        A.rgb[idx] = col[0];
        A.rgb[n + idx] = col[1];
        A.rgb[2 * n + idx] = col[2];
        if (A.state_end) {
#pragma unroll
            for (int c = 0; c < 8; c++) A.state_end[idx * 8 + c] = se[c];
        }
        if (A.lambda_end) A.lambda_end[idx] = lam;
        if (A.status) A.status[idx] = st.status;
        if (A.hit) A.hit[idx] = (uint8_t)hit;
        if (A.hit32) A.hit32[idx] = hit;
        if (A.n_accept) A.n_accept[idx] = st.nacc;
        if (A.n_reject) A.n_reject[idx] = st.nrej;
        ev = (st.status == RTGR_RAY_EVENT);
    }
    if (A.counters) {
        const unsigned long long c0 = wave_sum(valid ? 1ull : 0ull), c1 = wave_sum(st.nacc), c2 = wave_sum(st.nrej),
                                 c3 = wave_sum(st.nrhs), c4 = wave_sum(ev ? 1ull : 0ull),
                                 c5 = wave_sum(st.interior), c6 = wave_sum((valid && st.status >= RTGR_RAY_MAXSTEPS) ? 1ull : 0ull);
        if (lane == 0) {
            atomicAdd(&A.counters[0], c0);
            atomicAdd(&A.counters[1], c1);
            atomicAdd(&A.counters[2], c2);
            atomicAdd(&A.counters[3], c3);
            atomicAdd(&A.counters[4], c4);
            atomicAdd(&A.counters[5], c5);
            atomicAdd(&A.counters[6], c6);
        }
    }
}

template <class R, int METRIC, bool SPIN = false> constexpr int waves_per_simd_of(int mode) {
    if (METRIC >= RTGR_GENERIC_BASE) return sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD_GENERIC : RTGR_WAVES_PER_SIMD_GENERIC_F32;
    if (mode == MODE_FAR) return sizeof(R) == 8 ? (SPIN ? RTGR_WAVES_PER_SIMD_SPIN_FAR : RTGR_WAVES_PER_SIMD_FAR) : 4;
    return sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD : RTGR_WAVES_PER_SIMD_F32;
}

template <class R, int METRIC, bool SPIN>
static int launch_integrate(LaunchEnv& E, const IntegrateArgs<R>& IA, bool npts10, bool split, uint64_t waves, hipStream_t st) {
    constexpr bool USER = (METRIC == RTGR_GENERIC_BASE + RTGR_USER);
    DeviceCtx& D = E.d;
    const Knobs& K = E.knobs ? *E.knobs : D.knobs;
    auto grid = [&](int per_simd) {
        const uint64_t per_cu = K.waves_per_cu > 0 ? (uint64_t)K.waves_per_cu : (uint64_t)(4 * per_simd);
        uint64_t resident = (uint64_t)D.num_cu * per_cu;
        if (K.max_waves > 0 && (uint64_t)K.max_waves < resident) resident = (uint64_t)K.max_waves;
        return dim3((unsigned)(waves < resident ? waves : resident));
    };
    if (npts10 && split) {
        // round 0: FAR over the camera rays, NEAR over what it hands over (the NEAR pass keeps a ray to its end).
        // rounds = 2 (experiment, measured SLOWER: 9.9 -> 11.8 ms at 1024², 99.3 -> 101.9 ms at 4096²): the NEAR pass
        // hands rays that have left every object's reach back, and a second FAR + NEAR round carries them on — the
        // extra passes' own start-up and tails cost more than the NEAR tail they remove.  Each pass has its own queue
        // head (ctrl[0..7]).
        // … EXCEPT for long object lists (round 6): with tens of small objects along its path a ray is handed to the NEAR pass early and
        // would stay there — two waves per SIMD, every step a candidate for the scan — for the rest of its life; a second round takes
        // it back once it has left every reach.  Measured on example2 + N − 3 small spheres at 2048² (profiles/r06/
        // objects_cost_rounds_sweep.log; one / two rounds): 24 objects 41.9 / 42.1 ms, 32: 53.6 / 48.1, 48: 72.5 / 59.3, 64: 82.7 / 64.1,
        // 96: 102.4 / 88.0, 128: 88.1 / 83.8, 192: 124.1 / 123.0, 256: 169.2 / 187.7 (every ray is near something all the time); with
        // a = 0.8: 32: 50.0 / 52.2, 64: 59.0 / 58.8, 128: 95.4 / 83.0.  Automatic: two rounds for lists of 32 to 199 objects — a rule
        // read off one family of scenes; option `rounds` overrides it either way, and no choice changes a result bit.
        int rounds = K.rounds > 0 ? (int)K.rounds : ((IA.sc.nobj >= 32u && IA.sc.nobj < 200u) ? 2 : 1);
        rounds = rounds < 1 ? 1 : (rounds > 3 ? 3 : rounds);  // 2 queue heads per round; slot 6 is the early-list cursor
        IntegrateArgs<R> P = IA;
        for (int r = 0; r < rounds; r++) {
            P.ctrl = IA.ctrl + 2 * r;
            P.queue_chunk = IA.queue_chunk;
            if (r > 0) P.early = nullptr;  // the early list is round 0's
            P.pick_flag = r == 0 ? 0u : META_HANDBACK;
            if (r > 0) P.order = nullptr;
#ifdef RTGR_ROOT_STATS
            P.dbg = K.dbg_pass_far ? D.dbg : nullptr;
#endif
            { KernelTimer tm(D, st, 1);
              if constexpr (USER) {
                  const int fw = E.user->far_waves ? (int)E.user->far_waves : waves_per_simd_of<R, METRIC>(MODE_FAR);
                  HIP_TRY(launch_module(E.user->far, grid(fw).x, 64, st, P));
              } else {
                  bool four = false;
                  if constexpr (sizeof(R) == 8 && !SPIN && METRIC < RTGR_GENERIC_BASE && METRIC != RTGR_MINKOWSKI) {
                      // >= 24 rays per lane of a 4-waves/SIMD grid (6.3 M rays): see integrate_far4_kernel.  Measured
                      // 3 vs 4 waves: 4.2 M rays 22.3 / 22.5 ms, 8.4 M 44.1 / 43.3, 12.2 M 63.4 / 62.6, 16.8 M 85.6 / 84.5.
                      four = K.far4 >= 0 ? K.far4 != 0 : P.n >= (uint64_t)D.num_cu * 16 * 64 * 24;
                      if (four) hipLaunchKernelGGL((integrate_far4_kernel<R, METRIC>), grid(4), dim3(64), 0, st, P);
                  }
                  if (!four) hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_FAR>),
                                                grid(waves_per_simd_of<R, METRIC, SPIN>(MODE_FAR)), dim3(64), 0, st, P);
              } }
            P.pick_flag = META_HANDED;
            P.allow_handback = (r + 1 < rounds) ? 1u : 0u;
            P.handback_after = (uint32_t)(K.handback_after > 0 ? K.handback_after : 0);
            {   // The NEAR pass visits EVERY ray id, and its rays last ~6 steps: all its waves pop at the same time and
                // keep popping, so the device-scope atomic on the queue head (~90 M/s on one word) is what bounds it when
                // the chunks are small — measured 0.8 ms for 2.1 M rays at 42 ids per pop (50 k atomics), also for a
                // hand-back pass that picks up almost nothing.  So: 1/8 of a wave's share per pop, within [64, 256]
                // (sweep at 2.1 / 4.2 / 16.8 M rays: best at 64-128 / 128-256 / 256; 1024 parks stragglers: +30 %).
                const uint64_t share = P.n / ((uint64_t)grid(waves_per_simd_of<R, METRIC>(MODE_NEAR)).x * 8 + 1);
                const uint64_t nc = share < 64 ? 64 : (share > RTGR_QUEUE_CHUNK ? RTGR_QUEUE_CHUNK : share);
                P.queue_chunk = (uint32_t)(K.qchunk_near > 0 ? K.qchunk_near : (long)nc);
            }
#ifdef RTGR_ROOT_STATS
            P.dbg = K.dbg_pass_far ? nullptr : D.dbg;
#endif
            { KernelTimer tm(D, st, 3);
              // A small launch's NEAR pass ends on its longest-staying rays (one lane each, up to 370 steps in the
              // a = 0.8 scene), and such a wave steps faster alone on its SIMD than next to a second wave: ONE wave per
              // SIMD below 2.4 M rays, below 6.3 M with spin (1024²: a = 0.8 NEAR 2.17 -> 1.43 ms, a = 0 0.87 -> 0.64 ms;
              // same-run A/B of the frame time, tools/launch_ab.py ab: a = 0.8 at 2.1 M rays 18.64 -> 17.87 ms, 4.2 M
              // 33.67 -> 33.25, 8.4 M 63.85 -> 64.25; a = 0 at 2.1 M 12.43 -> 12.37, 4.2 M 23.31 -> 23.48: beyond these
              // sizes the second wave's throughput is worth more).
              int near_waves = waves_per_simd_of<R, METRIC>(MODE_NEAR);
              if constexpr (USER) if (E.user->near_waves) near_waves = (int)E.user->near_waves;
              dim3 gn = grid(near_waves);
              bool spin_rt = SPIN;
              if constexpr (USER) spin_rt = E.user->spin;   // (a unit's kernels may be a built-in metric's a = 0 instantiation)
              const uint64_t one_wave_below = (uint64_t)D.num_cu * 12 * 64 * (spin_rt ? 32 : 12);
              const long wn = K.waves_per_cu_near >= 0 ? K.waves_per_cu_near : (P.n < one_wave_below ? 4 : 0);
              if (wn > 0 && (uint64_t)D.num_cu * (uint64_t)wn < gn.x) gn.x = (unsigned)((uint64_t)D.num_cu * (uint64_t)wn);
              if constexpr (USER) HIP_TRY(launch_module(E.user->near, gn.x, 64, st, P));
              else hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_NEAR>), gn, dim3(64), 0, st, P);
            }
        }
    } else {
        IntegrateArgs<R> P = IA;
#ifdef RTGR_ROOT_STATS
        P.dbg = D.dbg;   // debug builds: the FULL pass reports its wave timeline too
#endif
        std::unique_ptr<KernelTimer> tm(new KernelTimer(D, st, 1));
        int full_waves = waves_per_simd_of<R, METRIC>(MODE_FULL);
        if constexpr (USER) {   // the occupancy the unit's FULL passes were built for (rtgr_user_near_waves / _f32_waves)
            const unsigned uw = sizeof(R) == 8 ? E.user->near_waves : E.user->f32_waves;
            if (uw) full_waves = (int)uw;
        }
        const dim3 g = grid(full_waves);
        // Float32, closed-form RHS, the reference's 10 sample points: the two-rays-per-lane kernel (rtgr_packed_f32.hpp); a wave is a
        // pool of 128 ray slots there.  It executes 1.55x fewer instructions per ray-step (PMC) but needs 205 registers — two waves
        // per SIMD, where a wave issues at most 83 % of the slots (tools/micro/valu_rates.hip: 3.75 vs 3.13 ticks per instruction at 2
        // vs 3 waves) — and v_pk_fma_f32 issues at 0.88 of a scalar v_fma_f32's rate: measured at 2048², +8 % with spin (the RHS, all
        // of which packs, is 40 % of the step), -5 % at a = 0 (a 50-flop RHS).  Hence automatic for a != 0 only; option pack = 0 / 1
        // forces the scalar / the packed kernel (A/B, tests).
        if constexpr (sizeof(R) == 4 && METRIC < RTGR_GENERIC_BASE) {
            if (npts10 && K.packfar > 0) {
                // EXPERIMENT (option packfar = 1; DESIGN §4.2a has the measurement): packed scan-free FAR pass at three waves per
                // SIMD, then the scalar NEAR pass over what it hands over
                const uint64_t waves2 = (IA.n + 127) / 128;
                const uint64_t per_cu = K.waves_per_cu > 0 ? (uint64_t)K.waves_per_cu : (uint64_t)(4 * RTGR_WAVES_PER_SIMD_PACKED_FAR);
                const uint64_t resident = (uint64_t)D.num_cu * per_cu;
                P.early = nullptr;          // (no early list: Float32 rays stay ~2 steps in the NEAR pass)
                P.pick_flag = 0;
                hipLaunchKernelGGL((integrate2_far_kernel<METRIC, SPIN>), dim3((unsigned)(waves2 < resident ? waves2 : resident)), dim3(64), 0, st, P);
                tm.reset();                                  // the FAR pass's time ends here …
                tm.reset(new KernelTimer(D, st, 3));         // … and the NEAR pass is timed as such
                P.pick_flag = META_HANDED;
                P.allow_handback = 0;
                const uint64_t share = P.n / ((uint64_t)g.x * 8 + 1);
                P.queue_chunk = (uint32_t)(K.qchunk_near > 0 ? K.qchunk_near : (long)(share < 64 ? 64 : (share > RTGR_QUEUE_CHUNK ? RTGR_QUEUE_CHUNK : share)));
                hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_NEAR>), g, dim3(64), 0, st, P);
                return RTGR_OK;
            }
            if (npts10 && (K.pack >= 0 ? K.pack != 0 : SPIN)) {
                const uint64_t waves2 = (IA.n + 127) / 128;
                const uint64_t per_cu = K.waves_per_cu > 0 ? (uint64_t)K.waves_per_cu : (uint64_t)(4 * RTGR_WAVES_PER_SIMD_PACKED);
                const uint64_t resident = (uint64_t)D.num_cu * per_cu;
                hipLaunchKernelGGL((integrate2_kernel<METRIC, SPIN>), dim3((unsigned)(waves2 < resident ? waves2 : resident)), dim3(64), 0, st, P);
                return RTGR_OK;
            }
        }
        if constexpr (USER) {
            hipFunction_t f = sizeof(R) == 8 ? (npts10 ? E.user->full10 : E.user->fulln)
                                             : (npts10 ? E.user->full10_f32 : E.user->fulln_f32);
            HIP_TRY(launch_module(f, g.x, 64, st, P));
        } else if (npts10) hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_FULL>), g, dim3(64), 0, st, P);
        else hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, false, MODE_FULL>), g, dim3(64), 0, st, P);
    }
    return RTGR_OK;
}

template <class R, int METRIC, bool SPIN>
static int launch_trace(LaunchEnv& E, const TraceArgs<R>& A, hipStream_t st) {
    constexpr bool USER = (METRIC == RTGR_GENERIC_BASE + RTGR_USER);
    DeviceCtx& D = E.d;
    const Knobs& K = E.knobs ? *E.knobs : D.knobs;
    if constexpr (METRIC < RTGR_GENERIC_BASE) if (K.tile) {
        const uint64_t tiles = ((A.ni + 7) / 8) * ((A.nrows + 7) / 8);
        const uint64_t blocks = (tiles + 3) / 4;
        if (E.after_setup) HIP_TRY(hipEventRecord(E.after_setup, st));
        KernelTimer tm(D, st, 1);
        hipLaunchKernelGGL((trace_kernel<R, METRIC, SPIN>), dim3((unsigned)blocks), dim3(256), 0, st, A);
        return RTGR_OK;
    }
    const uint64_t n = A.ni * A.nrows;
    const bool with_state = A.state_end != nullptr;
    const uint64_t chunk = pick_chunk<R>(D, E.ss, n, with_state);
    int rc = ensure_workspace(D, E.ss, workspace_bytes<R>(chunk, with_state), st);
    if (rc) return rc;
    const int recw = with_state ? REC_TAIL_STATE : REC_TAIL;   // (the event records' first HAND_W scalars overlay the hand-over records)
    char* base = (char*)E.ss.ws;
    R* rec = (R*)base;
    char* cur = base + align256(chunk * recw * sizeof(R));
    R* hand = (R*)cur;
    cur += align256(chunk * HAND_W * sizeof(R));
    uint32_t* meta = (uint32_t*)cur;
    cur += align256(chunk * 3 * sizeof(uint32_t));
    uint32_t* order = (uint32_t*)cur;
    cur += align256(chunk * sizeof(uint32_t));
    uint32_t* early = (uint32_t*)cur;
    cur += align256(chunk * sizeof(uint32_t));
    uint8_t* keys = (uint8_t*)cur;
    cur += align256(chunk);
    uint32_t* hist = (uint32_t*)cur;  // 256 bins + 256 running offsets
    cur += 4096;
    // Float32 rays last ~20 steps: one FULL pass wins (measured 5-6 %) — with a short list.  The FULL pass pays 9 distances per object
    // and accepted step, the FAR pass one reach test (and, with groups, not even that): from RTGR_F32_SPLIT_FROM objects on Float32
    // runs FAR + NEAR too (2048², 64 objects: a = 0 17.6 -> 11.1 ms, a = 0.8 14.9 -> 13.6; 256 objects 49.0 -> 22.1, 41.1 -> 25.7:
    // profiles/r06/f32_objects_split.log)
    // (a user unit carries Float32 twins of the FULL pass only)
    // (… and a unit whose objects come without a reach bound has nothing to decide a hand-over by: every accepted step is
    //  scanned, as the reference does — the single FULL pass; include/rtgr.h "user objects")
    bool split = (K.split >= 0 ? K.split != 0 : (sizeof(R) == 8 || A.sc.nobj >= (uint32_t)RTGR_F32_SPLIT_FROM)) && !(USER && sizeof(R) == 4);
    if constexpr (USER) if (E.user->has_objects && !E.user->has_reach) split = false;
    for (uint64_t off = 0; off < n; off += chunk) {
        const uint64_t m = (n - off) < chunk ? (n - off) : chunk;
        const R* s0 = A.state0 ? A.state0 + off * 8 : nullptr;  // null: prepare_kernel generates the camera rays
        // longest-expected-first queue order: pays off when a lane gets few rays (see rtgr_persistent.hpp).  Float32 rays
        // last ~20 steps, so above 2 M rays the order's own kernels and the scattered record traffic cost more than the
        // shorter tail returns (measured: 2048² a = 0.8 3.21 -> 2.98 ms, 4096² 12.4 -> 10.8 ms; 1024² a = 0 0.98 <- 1.56)
        const bool order_auto = sizeof(R) == 8 || m <= (1ull << 21);
        const bool use_order = METRIC != RTGR_MINKOWSKI && A.sc.metric != RTGR_MINKOWSKI && m >= 4096 && (K.order >= 0 ? K.order != 0 : order_auto);
        unsigned long long* q = E.ss.queue;  // one slot per stream: the stream orders this chunk behind the previous one
        hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, st, q, use_order ? hist : (uint32_t*)nullptr);
        IntegrateArgs<R> IA;
        std::memset(&IA, 0, sizeof IA);
        IA.sc = A.sc; IA.opt = A.opt; IA.state0 = s0; IA.order = use_order ? order : nullptr; IA.n = m; IA.rec = rec; IA.meta = meta; IA.recw = recw;
        IA.hand = hand; IA.ctrl = q; IA.counters = A.counters; IA.pick_flag = 0; IA.allow_handback = 0;
        IA.cam = A.cam; IA.ni = A.ni; IA.nj = A.nj; IA.j0 = A.j0; IA.jstride = A.jstride; IA.first = off;
        IA.keys = use_order ? keys : nullptr; IA.hist = hist; IA.nan_flag = A.nan_flag;
        IA.early = early; IA.near_early = (uint32_t)(K.near_early < 0 ? 64 : K.near_early);  // 0: no early list
        if (IA.near_early == 0u) IA.early = nullptr;
        // priority rotation among the waves of a SIMD: pays when a lane gets only a few rays (see rtgr_persistent.hpp)
        // measured FAR pass, off / on: 0.26 M rays 2.70 / 3.17 ms, 0.52 M 3.83 / 4.35, 1.05 M 7.05 / 6.29, 1.44 M 8.83 / 8.30,
        // 2.1 M 11.59 / 11.59, 16.8 M 83.8 / 84.8 -> on for 4..9 rays per lane of the 3-waves/SIMD grid
        IA.n_simd = (uint32_t)D.num_cu * 4u;
        const uint64_t lanes3 = (uint64_t)D.num_cu * 12 * 64;
        IA.fair_shift = (uint32_t)(K.fair >= 0 ? K.fair : ((m >= 4 * lanes3 && m < 9 * lanes3) ? 13 : 0));
        {   // ids per queue atomic, FAR / FULL pass: 1/16 of a wave's share of the job within [8, 16].  These rays last ~200
            // steps, so even 16.8 M rays in pops of 16 are 1 M atomics in 80 ms (the queue word takes ~90 M/s), and small
            // pops keep the end of the pass balanced: a wave never sits on ids another, idle wave could run.  Sweep of
            // 8..256 at 1 / 2.1 / 4.2 / 8.4 / 16.8 M rays (ms per frame, 256-capped rule before -> 16): a = 0 6.71 -> 6.66,
            // 12.26 -> 12.18, 23.17 -> 22.87, 45.40 -> 44.64, 88.45 -> 88.13; a = 0.8 10.61 -> 10.63, 19.00 -> 18.39,
            // 34.26 -> 33.03, 66.44 -> 64.06, 127.1 -> 125.0.
            // Float32 rays last ~22 steps (tol = eps^(3/4) = 6.4e-6): 4.2 M of them in pops of 16 ARE the atomic limit
            // (2048²: 3.5 -> 4.75 ms), so Float32 keeps the wide rule, up to RTGR_QUEUE_CHUNK ids per pop.
            const uint64_t per_wave = m / ((uint64_t)D.num_cu * 12 + 1);
            const uint64_t cap = sizeof(R) == 8 ? 16 : RTGR_QUEUE_CHUNK;
            uint64_t qc = per_wave / 16;
            qc = qc < 8 ? 8 : (qc > cap ? cap : qc);
            IA.queue_chunk = (uint32_t)(K.qchunk > 0 ? K.qchunk : (long)qc);
        }
        {   // ray set-up: camera ray (or the caller's state), ordering key, u̇(y0), initial dt, event sign -> start records
            KernelTimer tm(D, st, 0);
            if constexpr (USER) {
                HIP_TRY(launch_module(sizeof(R) == 8 ? E.user->prepare : E.user->prepare_f32, (unsigned)((m + 255) / 256), 256, st, IA));
            } else hipLaunchKernelGGL((prepare_kernel<R, METRIC, SPIN>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, IA);
            if (use_order) {
                hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(256), 0, st, hist, hist + 256);
                hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, keys, m, hist + 256, order);
            }
        }
        if (E.after_setup && off == 0) HIP_TRY(hipEventRecord(E.after_setup, st));
        rc = launch_integrate<R, METRIC, SPIN>(E, IA, A.opt.interp_points == 10, split, (m + 63) / 64, st);
        if (rc) return rc;
        ResolveArgs<R> RA;
        RA.sc = A.sc; RA.opt = A.opt; RA.rec = rec; RA.hand = hand; RA.meta = meta; RA.recw = recw; RA.select = K.groups ? 1u : 0u; RA.n = m; RA.offset = A.out_offset + off;
        RA.n_slab = A.plane_stride ? A.plane_stride : n; RA.rgb = A.rgb; RA.state_end = A.state_end; RA.lambda_end = A.lambda_end;
        RA.status = A.status; RA.hit = A.hit; RA.hit32 = A.hit32; RA.n_accept = A.n_accept; RA.n_reject = A.n_reject;
        {
            KernelTimer tm(D, st, 2);
            bool done = false;
            if constexpr (USER) if (E.user->has_objects) {   // the unit's resolve kernel: its objects' distance / objcolor
                HIP_TRY(launch_module(sizeof(R) == 8 ? E.user->resolve : E.user->resolve_f32, (unsigned)((m + 255) / 256), 256, st, RA));
                done = true;
            }
            if (!done) hipLaunchKernelGGL(resolve_kernel<R>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, RA);
        }
    }
    return RTGR_OK;
}

}  // namespace rtgr
