// rtgr_hip.hip — HIP kernels + the C ABI of include/rtgr.h  (gfx950 only; no CPU fallback, no compatibility paths).
//
// Kernel inventory
//   integrate_kernel / resolve_kernel (rtgr_persistent.hpp)   THE HOT PATH: persistent waves with lane refill,
//                                 then one thread per ray for root-find + colouring
//   trace_kernel<R,METRIC,SPIN>   simple variant (RTGR_KERNEL=tile): one lane per ray, 8x8-pixel tile per wave, whole
//                                 adaptive loop + event finder + colouring inline — an independent formulation
//   canvas_kernel<R>              make_canvas (src/RayTraceGR.jl:457-478)
//   eval_metric_kernel / eval_geodesic_kernel   parity hooks for the reference's unit tests
//   quantize_kernel               N0f8 rounding + transposed image layout of save() (:566-575)
//   pixels_*_kernel               AoS Pixel{T} array <-> ray states / rgb (:446-450, :532)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtgr_persistent.hpp"

using namespace rtgr;

// ---------------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int g_device = -1;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(RTGR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// One lane = one ray.  A wave owns an 8x8 pixel tile (lock-step efficiency 0.90 vs 0.45 for 64 consecutive
// pixels); a 256-thread workgroup owns 4 horizontally adjacent tiles.
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
        const uint8_t hit = colour_pixel<R>(A.sc, A.opt, se, col);
        A.rgb[idx] = col[0];
        A.rgb[n + idx] = col[1];
        A.rgb[2 * n + idx] = col[2];
        if (A.state_end) {
#pragma unroll
            for (int c = 0; c < 8; c++) A.state_end[idx * 8 + c] = se[c];
        }
        if (A.lambda_end) A.lambda_end[idx] = lam;
        if (A.status) A.status[idx] = st.status;
        if (A.hit) A.hit[idx] = hit;
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

template <class R>
__global__ __launch_bounds__(256) void canvas_kernel(DevScene<R> sc, DevCamera<R> cam, uint64_t ni, uint64_t nj,
                                                     uint64_t j0, uint64_t jstride, uint64_t first, uint64_t count, R* state0) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= count) return;
    const uint64_t idx = first + w;  // linear index inside the slab: i + k * ni, image row j = j0 + k * jstride
    R s[8];
    make_pixel<R>(sc, cam, ni, nj, idx % ni, j0 + (idx / ni) * jstride, s);
#pragma unroll
    for (int c = 0; c < 8; c++) state0[w * 8 + c] = s[c];
}

__global__ __launch_bounds__(256) void eval_metric_kernel(DevScene<double> sc, const double* x, uint64_t n, double* g,
                                                          double* dg, double* Gam) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double xx[4] = {x[4 * p], x[4 * p + 1], x[4 * p + 2], x[4 * p + 3]};
    double gg[4][4], dd[4][4][4];
    dmetric_dev<double>(sc.metric, sc.M, sc.a, xx, gg, dd);
    if (g) for (int q = 0; q < 16; q++) g[16 * p + q] = (&gg[0][0])[q];
    if (dg) for (int q = 0; q < 64; q++) dg[64 * p + q] = (&dd[0][0][0])[q];
    if (Gam) {
        double GG[4][4][4];
        christoffel_dev<double>(gg, dd, GG);
        for (int q = 0; q < 64; q++) Gam[64 * p + q] = (&GG[0][0][0])[q];
    }
}

template <int METRIC, bool SPIN>
RTGR_DEV void rhs_dispatch1(const double* s, double M, double a, double* ds) { rhs<double, METRIC, SPIN>(s, M, a, ds); }

__global__ __launch_bounds__(256) void eval_geodesic_kernel(DevScene<double> sc, const double* s, uint64_t n, int path,
                                                            double* ds) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double si[8], so[8];
    for (int c = 0; c < 8; c++) si[c] = s[8 * p + c];
    if (path == 1) {
        generic_rhs<double>(sc.metric, sc.M, sc.a, si, so);
    } else {
        const bool spin = sc.a != 0.0;
        if (sc.metric == RTGR_MINKOWSKI) rhs<double, RTGR_MINKOWSKI, false>(si, sc.M, sc.a, so);
        else if (sc.metric == RTGR_KS_REF) {
            if (spin) rhs<double, RTGR_KS_REF, true>(si, sc.M, sc.a, so);
            else rhs<double, RTGR_KS_REF, false>(si, sc.M, sc.a, so);
        } else {
            if (spin) rhs<double, RTGR_KS_TRUE, true>(si, sc.M, sc.a, so);
            else rhs<double, RTGR_KS_TRUE, false>(si, sc.M, sc.a, so);
        }
    }
    for (int c = 0; c < 8; c++) ds[8 * p + c] = so[c];
}

// N0f8 quantisation (round(255 x), FixedPointNumbers) + transposed layout image[j][i][c]
__global__ __launch_bounds__(256) void quantize_kernel(const double* rgb, uint64_t ni, uint64_t nj, uint8_t* img) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = ni * nj;
    if (idx >= n) return;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double v = rgb[c * n + idx];
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        img[idx * 3 + c] = (uint8_t)__builtin_rint(v * 255.0);  // idx = i + j*ni  ==  row j, column i
    }
}

// Pixel{Float64} AoS (11 doubles: pos 4, normal 4, rgb 3; src/RayTraceGR.jl:446-450) -> ray states
__global__ __launch_bounds__(256) void pixels_in_kernel(const double* px, uint64_t n, double* state0) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
#pragma unroll
    for (int c = 0; c < 8; c++) state0[idx * 8 + c] = px[idx * 11 + c];
}
// Pixel{T}(p.pos, p.normal, col)  (:532)
__global__ __launch_bounds__(256) void pixels_out_kernel(const double* px_in, const double* rgb, uint64_t n, double* px_out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
#pragma unroll
    for (int c = 0; c < 8; c++) px_out[idx * 11 + c] = px_in[idx * 11 + c];
#pragma unroll
    for (int c = 0; c < 3; c++) px_out[idx * 11 + 8 + c] = rgb[c * n + idx];
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// run-time loaded metric (rtgr_user_metric_load): the metric-dependent kernels of the pipeline, from a code object
// built out of rtgr_user_unit.hip.in
struct UserModule {
    hipModule_t module = nullptr;
    hipFunction_t far = nullptr, near = nullptr, full10 = nullptr, fulln = nullptr, canvas = nullptr,
                  prepare = nullptr, eval_metric = nullptr, eval_geodesic = nullptr;
};
static UserModule g_user;
constexpr int RTGR_UM = RTGR_GENERIC_BASE + RTGR_USER;

// hipModuleLaunchKernel with the arguments given as C++ values
template <class... Args>
static hipError_t launch_module(hipFunction_t f, unsigned grid, unsigned block, hipStream_t st, Args... args) {
    void* params[] = {(void*)&args...};
    return hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, st, params, nullptr);
}

template <class R>
static int convert_scene(const rtgr_scene* s, DevScene<R>& d) {
    if (!s) return fail(RTGR_ERR_BAD_ARG, "scene is NULL");
    if ((s->metric & ~RTGR_METRIC_GENERIC) > RTGR_USER) return fail(RTGR_ERR_BAD_ARG, "unknown metric enum");
    if ((s->metric & ~RTGR_METRIC_GENERIC) == RTGR_USER) {
        if (!g_user.module) return fail(RTGR_ERR_BAD_ARG, "RTGR_USER: no user metric loaded (rtgr_user_metric_load)");
        if (sizeof(R) != 8) return fail(RTGR_ERR_BAD_ARG, "RTGR_USER metrics are compiled for Float64 only");
    }
    if (s->nobj > RTGR_MAX_OBJECTS) return fail(RTGR_ERR_BAD_ARG, "too many objects");
    std::memset(&d, 0, sizeof d);
    d.metric = s->metric & ~RTGR_METRIC_GENERIC;
    d.nobj = s->nobj;
    d.M = (R)s->M;
    d.a = (R)s->a;
    for (uint32_t o = 0; o < s->nobj; o++) {
        if (s->obj[o].kind < RTGR_PLANE || s->obj[o].kind > RTGR_DISK)
            return fail(RTGR_ERR_BAD_ARG, "unknown object kind (abstract Object has no distance)");
        d.obj[o].kind = s->obj[o].kind;
        for (int q = 0; q < 9; q++) d.obj[o].p[q] = (R)s->obj[o].p[q];
    }
    return RTGR_OK;
}
template <class R>
static int convert_solver(const rtgr_solver* s, DevSolver<R>& d) {
    if (!s) return fail(RTGR_ERR_BAD_ARG, "solver is NULL");
    if (!(s->reltol > 0) || !(s->abstol > 0)) return fail(RTGR_ERR_BAD_ARG, "tolerances must be positive");
    if (!(s->lambda1 > s->lambda0)) return fail(RTGR_ERR_BAD_ARG, "lambda1 must exceed lambda0");
    if (s->max_steps == 0) return fail(RTGR_ERR_BAD_ARG, "max_steps must be positive");
    d.reltol = (R)s->reltol;
    d.abstol = (R)s->abstol;
    d.lambda0 = (R)s->lambda0;
    d.lambda1 = (R)s->lambda1;
    d.hit_threshold = (R)s->hit_threshold;
    for (int c = 0; c < 3; c++) d.miss_rgb[c] = (R)s->miss_rgb[c];
    d.max_steps = s->max_steps;
    d.interp_points = s->interp_points;
    return RTGR_OK;
}
template <class R>
static void convert_camera(const rtgr_camera* c, DevCamera<R>& d) {
    for (int a = 0; a < 4; a++) {
        d.pos[a] = (R)c->pos[a];
        d.widthx[a] = (R)c->widthx[a];
        d.widthy[a] = (R)c->widthy[a];
        d.normal[a] = (R)c->normal[a];
    }
}

static int bind_device(int dev);
static int ensure_device() {
    (void)hipGetLastError();  // every entry point starts here: drop a stale error left by an earlier (or foreign) call,
                              // so that the hipGetLastError() after our launches reports our launches only
    if (g_device >= 0) return RTGR_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RTGR_ERR_NO_DEVICE, "no HIP device visible; librtgr_hip has no CPU fallback");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    return bind_device(dev);
}

// ---- launch policy ---------------------------------------------------------------------------------------------
// Default: prepare -> integrate FAR / NEAR (persistent waves) -> resolve.  RTGR_KERNEL=tile selects the simple tile-per-wave
// kernel (kept as an independent formulation for A/B and cross-checks).  Tunables (experiments only):
//   RTGR_WAVES_PER_CU  resident waves per CU of the integrate kernel (default 8 = 2 per SIMD)
//   RTGR_CHUNK         rays per pipeline chunk (default 2^26, less if memory is short); bounds the library-owned workspace
//   RTGR_SPLIT=0       one FULL integrate pass instead of the FAR + NEAR pair
//   RTGR_ORDER=0       keep the natural ray order (default: longest-expected-first, see rtgr_persistent.hpp)
//   RTGR_FAIR=s        time slice 2^s clocks of the priority rotation (0 = off; default 13 for 0.8-1.8 M rays, else off)
//   RTGR_NEAR_EARLY=n  accepted steps at hand-over below which a ray is put on the NEAR pass's early list (default 64)
//   RTGR_WAVES_PER_CU_NEAR=n  resident waves per CU of the NEAR pass (default 4 below 1.6 M rays, else 8)
//   RTGR_FAR4=0/1      force the 3- / 4-waves-per-SIMD instantiation of the a = 0 FAR pass (default: by launch size)
static int g_num_cu = 0;
static unsigned long long* g_queue_pool = nullptr;  // RTGR_QUEUE_SLOTS work-queue heads, one per launch in flight
static unsigned g_queue_next = 0;
constexpr unsigned RTGR_QUEUE_SLOTS = 256;
#ifdef RTGR_ROOT_STATS
static unsigned long long* g_dbg = nullptr;
#endif
static void* g_ws = nullptr;  // library-owned workspace (event records, per-ray meta, generated ray states)
static size_t g_ws_bytes = 0;

// optional per-kernel timing (bench.py's roofline leg): hipEvents around each kernel of the pipeline
struct TimedLaunch { hipEvent_t a, b; int which; };
static bool g_timing = false;
static std::vector<TimedLaunch> g_timed;
static std::vector<hipEvent_t> g_event_pool;
static hipEvent_t take_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct KernelTimer {  // RAII: records start now, stop at scope exit
    hipStream_t st; int which; hipEvent_t a = nullptr, b = nullptr;
    KernelTimer(hipStream_t s, int w) : st(s), which(w) {
        if (g_timing) { a = take_event(); b = take_event(); (void)hipEventRecord(a, st); }
    }
    ~KernelTimer() {
        if (a) { (void)hipEventRecord(b, st); g_timed.push_back({a, b, which}); }
    }
};

static int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}
static bool use_tile_kernel() {
    const char* v = std::getenv("RTGR_KERNEL");
    return v && std::strcmp(v, "tile") == 0;
}
// Rays per pipeline chunk.  Every chunk pays the tails of its passes once, so bigger is better (8192² in one chunk instead
// of four: 581 -> 565 ms) and 288 GB of HBM can afford it: up to 2^26 rays (33.7 GB of workspace at 501 B/ray), halved
// until the workspace fits into a quarter of the memory that is free when it has to be (re)allocated.
static uint64_t chunk_rays() {
    const char* v = std::getenv("RTGR_CHUNK");
    uint64_t c = (v && *v) ? std::strtoull(v, nullptr, 10) : (1ull << 26);
    return c < 64 ? 64 : c;
}
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
static int ensure_workspace(size_t bytes) {
    if (bytes <= g_ws_bytes) return RTGR_OK;
    if (g_ws) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(g_ws); g_ws = nullptr; g_ws_bytes = 0; }
    HIP_TRY(hipMalloc(&g_ws, bytes));
    g_ws_bytes = bytes;
    return RTGR_OK;
}
template <class R>
static size_t workspace_bytes(uint64_t rays, bool with_state);
// the chunk size for a job of n rays: all of it if the workspace for that is there or fits, else the largest power-of-two
// fraction of chunk_rays() that does
template <class R>
static uint64_t pick_chunk(uint64_t n, bool with_state) {
    uint64_t chunk = n < chunk_rays() ? n : chunk_rays();
    if (workspace_bytes<R>(chunk, with_state) <= g_ws_bytes) return chunk;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return chunk;
    const size_t budget = (free_b + g_ws_bytes) / 4;  // the old workspace is released before the new one is taken
    while (chunk > (1ull << 20) && workspace_bytes<R>(chunk, with_state) > budget) chunk = (chunk + 1) / 2;
    return chunk;
}
template <class R>
static size_t workspace_bytes(uint64_t rays, bool with_state) {
    const int recw = with_state ? REC_W_STATE : REC_W;
    return align256(rays * recw * sizeof(R)) + align256(rays * 3 * sizeof(uint32_t)) +
           align256(rays * HAND_W * sizeof(R)) + 2 * align256(rays * sizeof(uint32_t)) + align256(rays) + 4096;
}

template <class R, int METRIC, bool SPIN>
static int launch_integrate(const IntegrateArgs<R>& IA, bool npts10, bool split, uint64_t waves, hipStream_t st) {
    auto grid = [&](int per_simd) {
        if (METRIC >= RTGR_GENERIC_BASE) per_simd = RTGR_WAVES_PER_SIMD_GENERIC;
        const uint64_t resident = (uint64_t)g_num_cu * (uint64_t)env_int("RTGR_WAVES_PER_CU", 4 * per_simd);
        return dim3((unsigned)(waves < resident ? waves : resident));
    };
    if (npts10 && split) {
        // round 0: FAR over the camera rays, NEAR over what it hands over (the NEAR pass keeps a ray to its end).
        // RTGR_ROUNDS=2 (experiment, measured SLOWER: 9.9 -> 11.8 ms at 1024², 99.3 -> 101.9 ms at 4096²): the NEAR pass
        // hands rays that have left every object's reach back, and a second FAR + NEAR round carries them on — the
        // extra passes' own start-up and tails cost more than the NEAR tail they remove.  Each pass has its own queue
        // head (ctrl[0..7]).
        int rounds = env_int("RTGR_ROUNDS", 1);
        rounds = rounds < 1 ? 1 : (rounds > 3 ? 3 : rounds);  // 2 queue heads per round; slot 6 is the early-list cursor
        IntegrateArgs<R> P = IA;
        for (int r = 0; r < rounds; r++) {
            P.ctrl = IA.ctrl + 2 * r;
            P.queue_chunk = IA.queue_chunk;
            if (r > 0) P.early = nullptr;  // the early list is round 0's
            P.pick_flag = r == 0 ? 0u : META_HANDBACK;
            if (r > 0) P.order = nullptr;
#ifdef RTGR_ROOT_STATS
            { const char* dp = std::getenv("RTGR_DBG_PASS"); P.dbg = (dp && std::strcmp(dp, "far") == 0) ? g_dbg : nullptr; }
#endif
            { KernelTimer tm(st, 1);
              if constexpr (METRIC == RTGR_UM) HIP_TRY(launch_module(g_user.far, grid(0).x, 64, st, P));
              else {
                  bool four = false;
                  if constexpr (sizeof(R) == 8 && !SPIN && METRIC < RTGR_GENERIC_BASE && METRIC != RTGR_MINKOWSKI) {
                      // >= 24 rays per lane of a 4-waves/SIMD grid (6.3 M rays): see integrate_far4_kernel.  Measured
                      // 3 vs 4 waves: 4.2 M rays 22.3 / 22.5 ms, 8.4 M 44.1 / 43.3, 12.2 M 63.4 / 62.6, 16.8 M 85.6 / 84.5.
                      // (RTGR_FAR4=0/1 forces)
                      const int force = env_int("RTGR_FAR4", -1);
                      four = force >= 0 ? force != 0 : P.n >= (uint64_t)g_num_cu * 16 * 64 * 24;
                      if (four) hipLaunchKernelGGL((integrate_far4_kernel<R, METRIC>), grid(4), dim3(64), 0, st, P);
                  }
                  if (!four) hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_FAR>),
                                                grid(sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD_FAR : 4), dim3(64), 0, st, P);
              } }
            P.pick_flag = META_HANDED;
            P.allow_handback = (r + 1 < rounds) ? 1u : 0u;
            {   // The NEAR pass visits EVERY ray id, and its rays last ~6 steps: all its waves pop at the same time and
                // keep popping, so the device-scope atomic on the queue head (~90 M/s on one word) is what bounds it when
                // the chunks are small — measured 0.8 ms for 2.1 M rays at 42 ids per pop (50 k atomics), also for a
                // hand-back pass that picks up almost nothing.  So: 1/8 of a wave's share per pop, within [64, 256]
                // (sweep at 2.1 / 4.2 / 16.8 M rays: best at 64-128 / 128-256 / 256; 1024 parks stragglers: +30 %).
                const uint64_t share = P.n / ((uint64_t)grid(sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD : RTGR_WAVES_PER_SIMD_F32).x * 8 + 1);
                const uint64_t nc = share < 64 ? 64 : (share > RTGR_QUEUE_CHUNK ? RTGR_QUEUE_CHUNK : share);
                P.queue_chunk = (uint32_t)env_int("RTGR_QCHUNK_NEAR", (int)nc);
            }
#ifdef RTGR_ROOT_STATS
            { const char* dp = std::getenv("RTGR_DBG_PASS"); P.dbg = (dp && std::strcmp(dp, "far") == 0) ? nullptr : g_dbg; }
#endif
            { KernelTimer tm(st, 3);
              if constexpr (METRIC == RTGR_UM) HIP_TRY(launch_module(g_user.near, grid(0).x, 64, st, P));
              else {
                  // A small launch's NEAR pass ends on its longest-staying rays (one lane each, up to 370 steps in the
                  // a = 0.8 scene), and such a wave steps faster alone on its SIMD than next to a second wave: ONE wave per
                  // SIMD below 1.6 M rays (1024²: a = 0.8 NEAR 2.17 -> 1.43 ms, a = 0 0.87 -> 0.64 ms; from 2 M rays on
                  // the second wave's throughput is worth more).  RTGR_WAVES_PER_CU_NEAR overrides.
                  dim3 gn = grid(sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD : RTGR_WAVES_PER_SIMD_F32);
                  const int wn = env_int("RTGR_WAVES_PER_CU_NEAR", P.n < (uint64_t)g_num_cu * 12 * 64 * 8 ? 4 : 0);
                  if (wn > 0 && (uint64_t)g_num_cu * wn < gn.x) gn.x = (unsigned)((uint64_t)g_num_cu * wn);
                  hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_NEAR>), gn, dim3(64), 0, st, P);
              } }
        }
    } else {
#ifdef RTGR_ROOT_STATS
        IntegrateArgs<R> IAd = IA; IAd.dbg = g_dbg;   // debug builds: the FULL pass reports its wave timeline too
        const IntegrateArgs<R>& IA = IAd;
#endif
        KernelTimer tm(st, 1);
        if constexpr (METRIC == RTGR_UM) HIP_TRY(launch_module(npts10 ? g_user.full10 : g_user.fulln, grid(0).x, 64, st, IA));
        else if (npts10) hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, true, MODE_FULL>), grid(sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD : RTGR_WAVES_PER_SIMD_F32), dim3(64), 0, st, IA);
        else hipLaunchKernelGGL((integrate_kernel<R, METRIC, SPIN, false, MODE_FULL>), grid(sizeof(R) == 8 ? RTGR_WAVES_PER_SIMD : RTGR_WAVES_PER_SIMD_F32), dim3(64), 0, st, IA);
    }
    return RTGR_OK;
}

template <class R, int METRIC, bool SPIN>
static int launch_trace(const TraceArgs<R>& A, hipStream_t st) {
    if constexpr (METRIC < RTGR_GENERIC_BASE) if (use_tile_kernel()) {
        const uint64_t tiles = ((A.ni + 7) / 8) * ((A.nrows + 7) / 8);
        const uint64_t blocks = (tiles + 3) / 4;
        KernelTimer tm(st, 1);
        hipLaunchKernelGGL((trace_kernel<R, METRIC, SPIN>), dim3((unsigned)blocks), dim3(256), 0, st, A);
        return RTGR_OK;
    }
    const uint64_t n = A.ni * A.nrows;
    const bool with_state = A.state_end != nullptr;
    const uint64_t chunk = pick_chunk<R>(n, with_state);
    int rc = ensure_workspace(workspace_bytes<R>(chunk, with_state));
    if (rc) return rc;
    const int recw = with_state ? REC_W_STATE : REC_W;
    char* base = (char*)g_ws;
    R* rec = (R*)base;
    char* cur = base + align256(chunk * recw * sizeof(R));
    uint32_t* meta = (uint32_t*)cur;
    cur += align256(chunk * 3 * sizeof(uint32_t));
    R* hand = (R*)cur;
    cur += align256(chunk * HAND_W * sizeof(R));
    uint32_t* order = (uint32_t*)cur;
    cur += align256(chunk * sizeof(uint32_t));
    uint32_t* early = (uint32_t*)cur;
    cur += align256(chunk * sizeof(uint32_t));
    uint8_t* keys = (uint8_t*)cur;
    cur += align256(chunk);
    uint32_t* hist = (uint32_t*)cur;  // 256 bins + 256 running offsets
    cur += 4096;
    // longest-expected-first queue order: pays off when a lane gets few rays (see rtgr_persistent.hpp); RTGR_ORDER=0/1 forces
    const int order_mode = env_int("RTGR_ORDER", -1);
    const bool split = env_int("RTGR_SPLIT", sizeof(R) == 8 ? 1 : 0) != 0;  // Float32 rays last ~20 steps: one FULL pass wins (measured 5-6 %)
    for (uint64_t off = 0; off < n; off += chunk) {
        const uint64_t m = (n - off) < chunk ? (n - off) : chunk;
        const R* s0 = A.state0 ? A.state0 + off * 8 : nullptr;  // null: prepare_kernel generates the camera rays
        const bool use_order = METRIC != RTGR_MINKOWSKI && m >= 4096 &&
                               (order_mode != 0);
        unsigned long long* q = g_queue_pool + 8 * (g_queue_next++ % RTGR_QUEUE_SLOTS);
        hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, st, q, use_order ? hist : (uint32_t*)nullptr);
        IntegrateArgs<R> IA;
        IA.sc = A.sc; IA.opt = A.opt; IA.state0 = s0; IA.order = use_order ? order : nullptr; IA.n = m; IA.rec = rec; IA.meta = meta; IA.recw = recw;
        IA.hand = hand; IA.ctrl = q; IA.counters = A.counters; IA.pick_flag = 0; IA.allow_handback = 0;
        IA.cam = A.cam; IA.ni = A.ni; IA.nj = A.nj; IA.j0 = A.j0; IA.jstride = A.jstride; IA.first = off;
        IA.keys = use_order ? keys : nullptr; IA.hist = hist;
        IA.early = early; IA.near_early = (uint32_t)env_int("RTGR_NEAR_EARLY", 64);  // 0: no early list
        if (IA.near_early == 0u) IA.early = nullptr;
        // priority rotation among the waves of a SIMD: pays when a lane gets only a few rays (see rtgr_persistent.hpp)
        // measured FAR pass, off / on: 0.26 M rays 2.70 / 3.17 ms, 0.52 M 3.83 / 4.35, 1.05 M 7.05 / 6.29, 1.44 M 8.83 / 8.30,
        // 2.1 M 11.59 / 11.59, 16.8 M 83.8 / 84.8 -> on for 4..9 rays per lane of the 3-waves/SIMD grid
        IA.n_simd = (uint32_t)g_num_cu * 4u;
        const uint64_t lanes3 = (uint64_t)g_num_cu * 12 * 64;
        IA.fair_shift = (uint32_t)env_int("RTGR_FAIR", (m >= 4 * lanes3 && m < 9 * lanes3) ? 13 : 0);
#ifdef RTGR_ROOT_STATS
        IA.dbg = nullptr;  // set per pass in launch_integrate (RTGR_DBG_PASS = far | near, default near)
#endif
        {   // ids per queue atomic: ~1/16 of a wave's share of the job, within [8, RTGR_QUEUE_CHUNK]
            const uint64_t per_wave = m / ((uint64_t)g_num_cu * 12 + 1);
            uint64_t qc = per_wave / 16;
            qc = qc < 8 ? 8 : (qc > RTGR_QUEUE_CHUNK ? RTGR_QUEUE_CHUNK : qc);
            IA.queue_chunk = (uint32_t)env_int("RTGR_QCHUNK", (int)qc);
        }
        {   // ray set-up: camera ray (or the caller's state), ordering key, u̇(y0), initial dt, event sign -> start records
            KernelTimer tm(st, 0);
            if constexpr (METRIC == RTGR_UM) HIP_TRY(launch_module(g_user.prepare, (unsigned)((m + 255) / 256), 256, st, IA));
            else hipLaunchKernelGGL((prepare_kernel<R, METRIC, SPIN>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, IA);
            if (use_order) {
                hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(256), 0, st, hist, hist + 256);
                hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, keys, m, hist + 256, order);
            }
        }
        rc = launch_integrate<R, METRIC, SPIN>(IA, A.opt.interp_points == 10, split, (m + 63) / 64, st);
        if (rc) return rc;
        ResolveArgs<R> RA;
        RA.sc = A.sc; RA.opt = A.opt; RA.rec = rec; RA.meta = meta; RA.recw = recw; RA.n = m; RA.offset = off;
        RA.n_slab = n; RA.rgb = A.rgb; RA.state_end = A.state_end; RA.lambda_end = A.lambda_end;
        RA.status = A.status; RA.hit = A.hit; RA.n_accept = A.n_accept; RA.n_reject = A.n_reject;
        {
            KernelTimer tm(st, 2);
            hipLaunchKernelGGL(resolve_kernel<R>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, RA);
        }
    }
    return RTGR_OK;
}

static int bind_device(int dev) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(RTGR_ERR_NO_DEVICE, std::string("librtgr_hip is built for gfx950 only; device is ") + p.gcnArchName);
    if (g_user.module && g_device != dev) {  // a code object is loaded into one device's context
        (void)hipModuleUnload(g_user.module); g_user = UserModule{};
    }
    if (g_queue_pool && g_device != dev) {
        (void)hipFree(g_queue_pool); g_queue_pool = nullptr;
        if (g_ws) { (void)hipFree(g_ws); g_ws = nullptr; g_ws_bytes = 0; }
    }
    if (!g_queue_pool) {
        HIP_TRY(hipMalloc((void**)&g_queue_pool, 8 * RTGR_QUEUE_SLOTS * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(g_queue_pool, 0, 8 * RTGR_QUEUE_SLOTS * sizeof(unsigned long long)));
    }
    g_num_cu = p.multiProcessorCount;
    g_device = dev;
    return RTGR_OK;
}

// generic dual-number RHS (RTGR_METRIC_GENERIC): Float64, metric kind as a template value 100 + kind
static int launch_generic(const TraceArgs<double>& A, hipStream_t st) {
    if (A.sc.metric == RTGR_USER) return launch_trace<double, RTGR_UM, true>(A, st);
    if (A.sc.metric == RTGR_KS_REF) return launch_trace<double, RTGR_GENERIC_BASE + RTGR_KS_REF, true>(A, st);
    return launch_trace<double, RTGR_GENERIC_BASE + RTGR_KS_TRUE, true>(A, st);
}
static int launch_generic(const TraceArgs<float>&, hipStream_t) { return RTGR_ERR_BAD_ARG; }

template <class R>
static int trace_device(const rtgr_scene* scene, const rtgr_solver* opt, const R* d_state0, const rtgr_camera* cam,
                        uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* d_rgb, const rtgr_ray_outputs* out,
                        rtgr_counters* d_counters, void* stream, uint64_t jstride = 1, uint64_t nrows_strided = 0) {
    int rc = ensure_device();
    if (rc) return rc;
    TraceArgs<R> A;
    std::memset(&A, 0, sizeof A);
    if ((rc = convert_scene<R>(scene, A.sc))) return rc;
    if ((rc = convert_solver<R>(opt, A.opt))) return rc;
    if (!d_rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    uint64_t nrows = j1 - j0;
    if (jstride != 1 || nrows_strided != 0) {  // rows j0, j0+jstride, … (nrows_strided of them)
        if (jstride == 0 || nrows_strided == 0 || j0 >= nj || j0 + (nrows_strided - 1) * jstride >= nj)
            return fail(RTGR_ERR_BAD_ARG, "bad strided row range: need j0 + (nrows-1)*jstride < nj");
        nrows = nrows_strided;
    } else if (j1 <= j0 || j1 > nj) {
        return fail(RTGR_ERR_BAD_ARG, "bad canvas range: need 0 <= j0 < j1 <= nj, ni > 0");
    }
    if (ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "bad canvas range: need 0 <= j0 < j1 <= nj, ni > 0");
    if (!d_state0 && !cam) return fail(RTGR_ERR_BAD_ARG, "need state0 or a camera");
    if (ni * nrows > (1ull << 40)) return fail(RTGR_ERR_BAD_ARG, "canvas too large");
    if (cam) convert_camera<R>(cam, A.cam);
    A.state0 = d_state0;
    A.ni = ni; A.nj = nj; A.j0 = j0; A.nrows = nrows; A.jstride = jstride;
    A.rgb = d_rgb;
    if (out) {
        A.state_end = (R*)out->state_end;
        A.lambda_end = (R*)out->lambda_end;
        A.status = out->status;
        A.hit = out->hit;
        A.n_accept = out->n_accept;
        A.n_reject = out->n_reject;
    }
    A.counters = (unsigned long long*)d_counters;
    hipStream_t st = (hipStream_t)stream;
    const bool spin = scene->a != 0.0;
    const bool generic = ((scene->metric & RTGR_METRIC_GENERIC) != 0 && A.sc.metric != RTGR_MINKOWSKI) || A.sc.metric == RTGR_USER;
    if (generic) {
        if (sizeof(R) != 8) return fail(RTGR_ERR_BAD_ARG, "RTGR_METRIC_GENERIC is compiled for Float64 only");
        if (use_tile_kernel()) return fail(RTGR_ERR_BAD_ARG, "RTGR_METRIC_GENERIC needs the persistent pipeline");
        rc = launch_generic(A, st);
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
        return RTGR_OK;
    }
    switch (A.sc.metric) {
        case RTGR_MINKOWSKI: rc = launch_trace<R, RTGR_MINKOWSKI, false>(A, st); break;
        case RTGR_KS_REF:
            rc = spin ? launch_trace<R, RTGR_KS_REF, true>(A, st) : launch_trace<R, RTGR_KS_REF, false>(A, st);
            break;
        default:
            rc = spin ? launch_trace<R, RTGR_KS_TRUE, true>(A, st) : launch_trace<R, RTGR_KS_TRUE, false>(A, st);
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return RTGR_OK;
}

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (bytes == 0) return RTGR_OK;
        HIP_TRY(hipMalloc(&p, bytes));
        return RTGR_OK;
    }
};

template <class R>
static int trace_host(const rtgr_scene* scene, const rtgr_solver* opt, const R* state0, const rtgr_camera* cam,
                      uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* rgb, const rtgr_ray_outputs* out,
                      rtgr_counters* ctr) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    if (ni == 0 || nj == 0 || j1 <= j0 || j1 > nj) return fail(RTGR_ERR_BAD_ARG, "bad canvas range: need 0 <= j0 < j1 <= nj, ni > 0");
    if (!state0 && !cam) return fail(RTGR_ERR_BAD_ARG, "need state0 or a camera");
    const uint64_t n = ni * (j1 - j0);
    if (state0) {  // the reference asserts !isnan on every metric call (src/RayTraceGR.jl:279)
        for (uint64_t q = 0; q < n * 8; q++)
            if (state0[q] != state0[q]) return fail(RTGR_ERR_NAN_INPUT, "NaN in an input ray (AssertionError in the reference, :279)");
    }
    DevBuf b_s0, b_rgb, b_se, b_lam, b_st, b_hit, b_na, b_nr, b_ctr;
    if (state0) {
        if ((rc = b_s0.alloc(n * 8 * sizeof(R)))) return rc;
        HIP_TRY(hipMemcpy(b_s0.p, state0, n * 8 * sizeof(R), hipMemcpyHostToDevice));
    }
    if ((rc = b_rgb.alloc(n * 3 * sizeof(R)))) return rc;
    rtgr_ray_outputs dout;
    std::memset(&dout, 0, sizeof dout);
    if (out) {
        if (out->state_end) { if ((rc = b_se.alloc(n * 8 * sizeof(R)))) return rc; dout.state_end = b_se.p; }
        if (out->lambda_end) { if ((rc = b_lam.alloc(n * sizeof(R)))) return rc; dout.lambda_end = b_lam.p; }
        if (out->status) { if ((rc = b_st.alloc(n))) return rc; dout.status = (uint8_t*)b_st.p; }
        if (out->hit) { if ((rc = b_hit.alloc(n))) return rc; dout.hit = (uint8_t*)b_hit.p; }
        if (out->n_accept) { if ((rc = b_na.alloc(n * 4))) return rc; dout.n_accept = (uint32_t*)b_na.p; }
        if (out->n_reject) { if ((rc = b_nr.alloc(n * 4))) return rc; dout.n_reject = (uint32_t*)b_nr.p; }
    }
    if ((rc = b_ctr.alloc(sizeof(rtgr_counters)))) return rc;
    HIP_TRY(hipMemset(b_ctr.p, 0, sizeof(rtgr_counters)));
    rc = trace_device<R>(scene, opt, (const R*)b_s0.p, cam, ni, nj, j0, j1, (R*)b_rgb.p, &dout, (rtgr_counters*)b_ctr.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(rgb, b_rgb.p, n * 3 * sizeof(R), hipMemcpyDeviceToHost));
    if (out) {
        if (out->state_end) HIP_TRY(hipMemcpy(out->state_end, b_se.p, n * 8 * sizeof(R), hipMemcpyDeviceToHost));
        if (out->lambda_end) HIP_TRY(hipMemcpy(out->lambda_end, b_lam.p, n * sizeof(R), hipMemcpyDeviceToHost));
        if (out->status) HIP_TRY(hipMemcpy(out->status, b_st.p, n, hipMemcpyDeviceToHost));
        if (out->hit) HIP_TRY(hipMemcpy(out->hit, b_hit.p, n, hipMemcpyDeviceToHost));
        if (out->n_accept) HIP_TRY(hipMemcpy(out->n_accept, b_na.p, n * 4, hipMemcpyDeviceToHost));
        if (out->n_reject) HIP_TRY(hipMemcpy(out->n_reject, b_nr.p, n * 4, hipMemcpyDeviceToHost));
    }
    if (ctr) HIP_TRY(hipMemcpy(ctr, b_ctr.p, sizeof(rtgr_counters), hipMemcpyDeviceToHost));
    return RTGR_OK;
}

static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" {

int rtgr_abi_version(void) { return RTGR_ABI_VERSION; }
const char* rtgr_last_error(void) { return g_err.c_str(); }

int rtgr_init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RTGR_ERR_NO_DEVICE, "no HIP device visible; librtgr_hip has no CPU fallback");
    if (device >= n) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    return bind_device(dev);
}
int rtgr_shutdown(void) {
    if (g_user.module) { (void)hipDeviceSynchronize(); (void)hipModuleUnload(g_user.module); g_user = UserModule{}; }
    if (g_queue_pool) { (void)hipFree(g_queue_pool); g_queue_pool = nullptr; }
    if (g_ws) { (void)hipFree(g_ws); g_ws = nullptr; g_ws_bytes = 0; }
    g_device = -1;
    return RTGR_OK;
}
int rtgr_solver_defaults(rtgr_solver* s, int is_f32) {
    if (!s) return fail(RTGR_ERR_BAD_ARG, "solver is NULL");
    const double eps = is_f32 ? 1.1920928955078125e-07 : 2.220446049250313e-16;
    s->reltol = s->abstol = std::pow(eps, 0.75);  // eps(T)^(3/4)   src/RayTraceGR.jl:485
    s->lambda0 = 0.0;                             // :497
    s->lambda1 = 100.0;
    s->hit_threshold = 0.01;                      // :519
    s->miss_rgb[0] = 1.0;                         // :528
    s->miss_rgb[1] = s->miss_rgb[2] = 0.0;
    s->max_steps = 100000;
    s->interp_points = 10;
    return RTGR_OK;
}
int rtgr_device_info(char* name, uint64_t name_len, int* n_cu, int* clock_mhz, int* wavefront) {
    int rc = ensure_device();
    if (rc) return rc;
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, g_device));
    if (name && name_len) {
        std::snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (wavefront) *wavefront = p.warpSize;
    return RTGR_OK;
}

int rtgr_timing_enable(int on) {
    g_timing = on != 0;
    return RTGR_OK;
}
int rtgr_timing_read(double ms[4], uint64_t launches[4]) {
    if (!ms || !launches) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    for (int w = 0; w < 4; w++) { ms[w] = 0.0; launches[w] = 0; }
    for (auto& t : g_timed) {
        HIP_TRY(hipEventSynchronize(t.b));
        float e = 0.f;
        HIP_TRY(hipEventElapsedTime(&e, t.a, t.b));
        ms[t.which] += e;
        launches[t.which] += 1;
        g_event_pool.push_back(t.a);
        g_event_pool.push_back(t.b);
    }
    g_timed.clear();
    return RTGR_OK;
}

int rtgr_reserve_workspace(uint64_t n_rays, int with_state_end, int is_f32) {
    int rc = ensure_device();
    if (rc) return rc;
    if (is_f32) return ensure_workspace(workspace_bytes<float>(pick_chunk<float>(n_rays, with_state_end != 0), with_state_end != 0));
    return ensure_workspace(workspace_bytes<double>(pick_chunk<double>(n_rays, with_state_end != 0), with_state_end != 0));
}

int rtgr_trace_device_f64(const rtgr_scene* scene, const rtgr_solver* opt, const double* d_state0,
                          const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* d_rgb,
                          const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    return trace_device<double>(scene, opt, d_state0, cam, ni, nj, j0, j1, d_rgb, out, d_counters, stream);
}
int rtgr_trace_device_f32(const rtgr_scene* scene, const rtgr_solver* opt, const float* d_state0,
                          const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* d_rgb,
                          const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    return trace_device<float>(scene, opt, d_state0, cam, ni, nj, j0, j1, d_rgb, out, d_counters, stream);
}
int rtgr_trace_rows_device_f64(const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni,
                               uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, double* d_rgb,
                               const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    if (!cam) return fail(RTGR_ERR_BAD_ARG, "camera is NULL");
    return trace_device<double>(scene, opt, nullptr, cam, ni, nj, j0, j0 + 1, d_rgb, out, d_counters, stream, jstride, nrows);
}
int rtgr_trace_rows_device_f32(const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni,
                               uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, float* d_rgb,
                               const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    if (!cam) return fail(RTGR_ERR_BAD_ARG, "camera is NULL");
    return trace_device<float>(scene, opt, nullptr, cam, ni, nj, j0, j0 + 1, d_rgb, out, d_counters, stream, jstride, nrows);
}
int rtgr_trace_f64(const rtgr_scene* scene, const rtgr_solver* opt, const double* state0, const rtgr_camera* cam,
                   uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* rgb, const rtgr_ray_outputs* out,
                   rtgr_counters* ctr) {
    return trace_host<double>(scene, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr);
}
int rtgr_trace_f32(const rtgr_scene* scene, const rtgr_solver* opt, const float* state0, const rtgr_camera* cam,
                   uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* rgb, const rtgr_ray_outputs* out,
                   rtgr_counters* ctr) {
    return trace_host<float>(scene, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr);
}

int rtgr_trace_pixels_f64(const rtgr_scene* scene, const rtgr_solver* opt, const double* pixels_in, uint64_t ni,
                          uint64_t nj, double* pixels_out, rtgr_counters* ctr) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!pixels_in || !pixels_out) return fail(RTGR_ERR_BAD_ARG, "pixels is NULL");
    if (ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "empty canvas");
    const uint64_t n = ni * nj;
    for (uint64_t p = 0; p < n; p++)
        for (int c = 0; c < 8; c++)
            if (pixels_in[p * 11 + c] != pixels_in[p * 11 + c])
                return fail(RTGR_ERR_NAN_INPUT, "NaN in an input pixel (AssertionError in the reference, :279)");
    DevBuf b_px, b_s0, b_rgb, b_out, b_ctr;
    if ((rc = b_px.alloc(n * 11 * 8))) return rc;
    if ((rc = b_s0.alloc(n * 8 * 8))) return rc;
    if ((rc = b_rgb.alloc(n * 3 * 8))) return rc;
    if ((rc = b_out.alloc(n * 11 * 8))) return rc;
    if ((rc = b_ctr.alloc(sizeof(rtgr_counters)))) return rc;
    HIP_TRY(hipMemset(b_ctr.p, 0, sizeof(rtgr_counters)));
    HIP_TRY(hipMemcpy(b_px.p, pixels_in, n * 11 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pixels_in_kernel, dim3(nblk(n)), dim3(256), 0, nullptr, (const double*)b_px.p, n, (double*)b_s0.p);
    rc = trace_device<double>(scene, opt, (const double*)b_s0.p, nullptr, ni, nj, 0, nj, (double*)b_rgb.p, nullptr,
                              (rtgr_counters*)b_ctr.p, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(pixels_out_kernel, dim3(nblk(n)), dim3(256), 0, nullptr, (const double*)b_px.p,
                       (const double*)b_rgb.p, n, (double*)b_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(pixels_out, b_out.p, n * 11 * 8, hipMemcpyDeviceToHost));
    if (ctr) HIP_TRY(hipMemcpy(ctr, b_ctr.p, sizeof(rtgr_counters), hipMemcpyDeviceToHost));
    return RTGR_OK;
}

int rtgr_trace_one_f64(const rtgr_scene* scene, const rtgr_solver* opt, const double pos[4], const double normal[4],
                       double rgb[3], double state_end[8], uint8_t* status) {
    if (!pos || !normal || !rgb) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    double s0[8];
    for (int c = 0; c < 4; c++) { s0[c] = pos[c]; s0[4 + c] = normal[c]; }
    rtgr_ray_outputs out;
    std::memset(&out, 0, sizeof out);
    out.state_end = state_end;
    out.status = status;
    return trace_host<double>(scene, opt, s0, nullptr, 1, 1, 0, 1, rgb, &out, nullptr);
}

int rtgr_make_canvas_device_f64(const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                                uint64_t j1, double* d_state0, void* stream) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!cam || !d_state0) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (ni == 0 || nj == 0 || j1 <= j0 || j1 > nj) return fail(RTGR_ERR_BAD_ARG, "bad canvas range");
    DevScene<double> sc;
    if ((rc = convert_scene<double>(scene, sc))) return rc;
    DevCamera<double> c;
    convert_camera<double>(cam, c);
    const uint64_t n = ni * (j1 - j0);
    if (sc.metric == RTGR_USER)
        HIP_TRY(launch_module(g_user.canvas, nblk(n), 256, (hipStream_t)stream, sc, c, ni, nj, j0, (uint64_t)1, (uint64_t)0,
                              n, d_state0));
    else
        hipLaunchKernelGGL(canvas_kernel<double>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, sc, c, ni, nj, j0,
                           (uint64_t)1, (uint64_t)0, n, d_state0);
    HIP_TRY(hipGetLastError());
    return RTGR_OK;
}
int rtgr_make_canvas_f64(const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                         uint64_t j1, double* state0) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!state0) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (ni == 0 || nj == 0 || j1 <= j0 || j1 > nj) return fail(RTGR_ERR_BAD_ARG, "bad canvas range");
    const uint64_t n = ni * (j1 - j0);
    DevBuf b;
    if ((rc = b.alloc(n * 64))) return rc;
    if ((rc = rtgr_make_canvas_device_f64(scene, cam, ni, nj, j0, j1, (double*)b.p, nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(state0, b.p, n * 64, hipMemcpyDeviceToHost));
    return RTGR_OK;
}

int rtgr_eval_metric_f64(const rtgr_scene* scene, const double* x, uint64_t n, double* g, double* dg, double* Gam) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!x) return fail(RTGR_ERR_BAD_ARG, "x is NULL");
    if (n == 0) return RTGR_OK;
    for (uint64_t q = 0; q < 4 * n; q++)
        if (x[q] != x[q]) return fail(RTGR_ERR_NAN_INPUT, "NaN coordinate (AssertionError in the reference, :279)");
    DevScene<double> sc;
    if ((rc = convert_scene<double>(scene, sc))) return rc;
    DevBuf bx, bg, bd, bG;
    if ((rc = bx.alloc(n * 32))) return rc;
    HIP_TRY(hipMemcpy(bx.p, x, n * 32, hipMemcpyHostToDevice));
    if (g && (rc = bg.alloc(n * 128))) return rc;
    if (dg && (rc = bd.alloc(n * 512))) return rc;
    if (Gam && (rc = bG.alloc(n * 512))) return rc;
    if (sc.metric == RTGR_USER)
        HIP_TRY(launch_module(g_user.eval_metric, nblk(n), 256, (hipStream_t) nullptr, sc, (const double*)bx.p, n,
                              (double*)bg.p, (double*)bd.p, (double*)bG.p));
    else
        hipLaunchKernelGGL(eval_metric_kernel, dim3(nblk(n)), dim3(256), 0, nullptr, sc, (const double*)bx.p, n,
                           (double*)bg.p, (double*)bd.p, (double*)bG.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (g) HIP_TRY(hipMemcpy(g, bg.p, n * 128, hipMemcpyDeviceToHost));
    if (dg) HIP_TRY(hipMemcpy(dg, bd.p, n * 512, hipMemcpyDeviceToHost));
    if (Gam) HIP_TRY(hipMemcpy(Gam, bG.p, n * 512, hipMemcpyDeviceToHost));
    return RTGR_OK;
}

int rtgr_eval_geodesic_f64(const rtgr_scene* scene, const double* s, uint64_t n, int path, double* ds) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!s || !ds) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (path != 0 && path != 1) return fail(RTGR_ERR_BAD_ARG, "path must be 0 (production) or 1 (generic duals)");
    if (n == 0) return RTGR_OK;
    for (uint64_t q = 0; q < 8 * n; q++)
        if (s[q] != s[q]) return fail(RTGR_ERR_NAN_INPUT, "NaN state (AssertionError in the reference, :279)");
    DevScene<double> sc;
    if ((rc = convert_scene<double>(scene, sc))) return rc;
    DevBuf bi, bo;
    if ((rc = bi.alloc(n * 64))) return rc;
    if ((rc = bo.alloc(n * 64))) return rc;
    HIP_TRY(hipMemcpy(bi.p, s, n * 64, hipMemcpyHostToDevice));
    if (sc.metric == RTGR_USER)  // a user metric has the generic path only
        HIP_TRY(launch_module(g_user.eval_geodesic, nblk(n), 256, (hipStream_t) nullptr, sc, (const double*)bi.p, n,
                              (double*)bo.p));
    else
        hipLaunchKernelGGL(eval_geodesic_kernel, dim3(nblk(n)), dim3(256), 0, nullptr, sc, (const double*)bi.p, n, path,
                           (double*)bo.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ds, bo.p, n * 64, hipMemcpyDeviceToHost));
    return RTGR_OK;
}

int rtgr_user_metric_unload(void) {
    if (!g_user.module) return RTGR_OK;
    HIP_TRY(hipDeviceSynchronize());  // kernels of the module may still be in flight
    hipModule_t m = g_user.module;
    g_user = UserModule{};
    HIP_TRY(hipModuleUnload(m));
    return RTGR_OK;
}

int rtgr_user_metric_load(const char* code_object_path) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!code_object_path || !*code_object_path) return fail(RTGR_ERR_BAD_ARG, "code object path is NULL or empty");
    UserModule u;
    hipError_t e = hipModuleLoad(&u.module, code_object_path);
    if (e != hipSuccess)
        return fail(RTGR_ERR_HIP, std::string("hipModuleLoad(") + code_object_path + "): " + hipGetErrorString(e));
    auto bail = [&](const std::string& why) {
        (void)hipModuleUnload(u.module);
        return fail(RTGR_ERR_BAD_ARG, std::string(code_object_path) + ": " + why);
    };
    {   // the unit must have been built against this library's headers
        hipDeviceptr_t dptr = nullptr;
        size_t bytes = 0;
        unsigned ver = 0;
        if (hipModuleGetGlobal(&dptr, &bytes, u.module, "rtgr_user_abi_version") != hipSuccess || bytes != sizeof ver)
            return bail("not a user-metric code object (no rtgr_user_abi_version)");
        if (hipMemcpyDtoH(&ver, dptr, sizeof ver) != hipSuccess) return bail("cannot read rtgr_user_abi_version");
        if (ver != RTGR_ABI_VERSION) return bail("built against another ABI version");
    }
    struct { hipFunction_t* f; const char* name; } want[] = {
        {&u.far, "rtgr_user_integrate_far"},       {&u.near, "rtgr_user_integrate_near"},
        {&u.full10, "rtgr_user_integrate_full10"}, {&u.fulln, "rtgr_user_integrate_fulln"},
        {&u.canvas, "rtgr_user_canvas"},           {&u.eval_metric, "rtgr_user_eval_metric"},
        {&u.eval_geodesic, "rtgr_user_eval_geodesic"}, {&u.prepare, "rtgr_user_prepare"}};
    for (auto& w : want)
        if (hipModuleGetFunction(w.f, u.module, w.name) != hipSuccess) return bail(std::string("missing kernel ") + w.name);
    if ((rc = rtgr_user_metric_unload())) { (void)hipModuleUnload(u.module); return rc; }
    g_user = u;
    return RTGR_OK;
}

int rtgr_user_metric_loaded(void) { return g_user.module ? 1 : 0; }

#ifdef RTGR_ROOT_STATS
// debug builds only: a device buffer the NEAR pass writes per-wave {start, end, iterations, rays} and per-ray stays into
// (tools/debug_near_waves.py)
int rtgr_debug_set_buffer(void* d_buf) { g_dbg = (unsigned long long*)d_buf; return RTGR_OK; }
// debug builds only (tools/debug_root_iters.py): copy the head of the library workspace (the event records) to the host
int rtgr_debug_workspace(void* dst, uint64_t bytes) {
    if (!g_ws || bytes > g_ws_bytes) return fail(RTGR_ERR_BAD_ARG, "no workspace / too many bytes");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dst, g_ws, bytes, hipMemcpyDeviceToHost));
    return RTGR_OK;
}
#endif

int rtgr_quantize_device_f64(const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, void* stream) {
    int rc = ensure_device();
    if (rc) return rc;
    if (!d_rgb || !d_img || ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "bad argument");
    hipLaunchKernelGGL(quantize_kernel, dim3(nblk(ni * nj)), dim3(256), 0, (hipStream_t)stream, d_rgb, ni, nj, d_img);
    HIP_TRY(hipGetLastError());
    return RTGR_OK;
}

}  // extern "C"
