// rtgr_host.hpp — host-side state of librtgr_hip.so: contexts, per-stream workspaces, launch policy knobs.
//
// SURVEY §8(b) "Ownership / Threading": the library owns device buffers and streams inside an OPAQUE CONTEXT
// (rtgr_context, include/rtgr.h); calls are re-entrant per context, one host thread may drive several devices and
// several host threads may drive one.  Layout of that state:
//
//   rtgr_context            the devices one host process drives (rtgr_create(device_ids, n))
//     DeviceCtx[n]          one per entry of device_ids (the same physical GPU may be listed twice: two "logical
//                           devices" with streams and workspaces of their own — how the multi-device path is tested on a
//                           one-GPU box)
//       StreamState[*]      one per hipStream_t the caller has used on that device: the pipeline workspace (start /
//                           hand-over / event records, per-ray meta, queue order, work-queue heads).  Launches on ONE
//                           stream are ordered by the stream and share a workspace; launches on DIFFERENT streams get
//                           different workspaces, so they never race (round 1 had one process-global workspace)
//       user modules        run-time compiled units (a metric, Object subtypes, or both), keyed by the 64-bit id a scene carries
//                           (rtgr_scene.user_metric)
//       staging             pinned host buffers + device buffers + 3 streams of the host-pointer entry points
//
// A workspace only grows; the superseded allocation is RETIRED, not freed (a hipGraph captured earlier may still
// reference it) until rtgr_destroy / rtgr_trim.  Growth during stream capture is refused (hipMalloc is not capturable):
// reserve first (rtgr_reserve_workspace).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rtgr.h"
#include "rtgr_args.hpp"

namespace rtgr {

int fail(int code, const std::string& msg);  // records the calling thread's last error, returns `code`
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return ::rtgr::fail(RTGR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

// ---- launch policy knobs ------------------------------------------------------------------------------------------------
// Parsed ONCE from the environment when a context is created (RTGR_<NAME>), changeable per context with
// rtgr_set_option(ctx, "name", value).  -1 = "auto" (the library decides from the launch size).  Experiments and
// schedule-invariance tests only: no knob changes a result bit, except `tile` and `pack`, which select another formulation of
// the same algorithm (results equal up to rounding).
struct Knobs {
    long waves_per_cu = -1;       // resident waves per CU of the integrate kernels (auto: 4 x waves/SIMD of the instantiation)
    long waves_per_cu_near = -1;  // ... of the NEAR pass (auto: 4 below 2.4 M rays, 6.3 M with spin)
    long chunk = -1;              // rays per pipeline chunk (auto: 2^26, less if memory is short)
    long split = -1;              // 0: one FULL pass instead of FAR + NEAR (auto: on for f64, off for f32)
    long order = -1;              // 0: natural ray order, 1: longest-expected-first (auto: ordered from 4096 rays; Float32 only up to 2 M)
    long fair = -1;               // log2 of the priority-rotation time slice in clocks, 0 = off (auto: by launch size)
    long near_early = 64;         // accepted steps at hand-over below which a ray goes on the NEAR pass's early list (0: no list)
    long far4 = -1;               // 0/1: force the 3- / 4-waves-per-SIMD a = 0 FAR instantiation (auto: by launch size)
    long rounds = -1;             // FAR / NEAR hand-back rounds (1..3); auto: 2 for object lists of 32..199, else 1 (rtgr_pipeline.hpp)
    long handback_after = 0;      // rounds > 1: the NEAR pass hands back only rays that have stayed this many accepted steps (experiment)
    long qchunk = -1;             // ray ids per queue atomic, FAR / FULL pass (auto)
    long qchunk_near = -1;        // ... NEAR pass (auto)
    long tile = 0;                // 1: the simple tile-per-wave kernel (RTGR_KERNEL=tile), an independent formulation
    long host_chunk = -1;         // host entry points: rays per H2D/compute/D2H pipeline piece (auto: 2^20)
    long dbg_pass_far = 0;        // debug builds: which pass reports its wave timeline
    long pack = -1;               // Float32 closed-form FULL pass: 0 = scalar kernel (one ray per lane), 1 = packed two-rays-per-lane
                                  // kernel, -1 = packed where it pays (a != 0)
    long packfar = 0;             // Float32 experiment: 1 = packed scan-free FAR pass at three waves per SIMD + the scalar NEAR pass
    long peer = -1;               // multi-device gather (rtgr_trace_sharded_device_*): 0 = always stage the rows through pinned
                                  // host memory (the no-peer-access fallback, forced), 1 = peer copies or fail, -1 = peer copies
                                  // where rtgr_create could enable peer access, the fallback elsewhere
    long max_waves = -1;          // cap of the integrate kernels' grid, in waves (the load-time probe: three waves over its 1024 rays make
                                  // every lane refill many times; results never depend on it)
    long unit_probe = 1;          // run-time units: trace a small probe frame through both pass structures at load and refuse a unit
                                  // whose frames are irreproducible or disagree (rtgr_units.hip: probe_unit); 0 = skip
    long unit_audit = 1;          // … and audit the code object for the compiler's EXEC-flip fault before loading it; 0 = skip
                                  // (a test hook: the probe must catch a faulty unit on its own)
    long groups = 1;              // object lists beyond the argument block: 1 = their spheres are sorted into groups of neighbours with a bounding
                                  // sphere each, which the FAR pass's reach test asks first (DevScene, rtgr_args.hpp); 0 = every object is asked; >= 2: on, with
                                  // that many spheres per group at most (experiments; 8 = RTGR_GROUP_MAX is the measured optimum)
    long scene_check = 1;         // scenes whose USER OBJECTS bring their own reach bound: the first trace of each (unit, object list, metric
                                  // parameters, solver, camera) runs rtgr_scene_check's comparison on a coarse sample of the call's own rays and
                                  // refuses a scene whose FAR + NEAR passes lose what the FULL pass finds (rtgr_units.hip: auto_scene_check); 0 = off
};
const char* const* knob_names();  // NULL-terminated
long* knob_slot(Knobs& k, const char* name);

// ---- run-time loaded unit: the kernels of the pipeline that depend on the caller's metric / objects, from a code object ------
struct UserModule {
    uint64_t id = 0;
    hipModule_t module = nullptr;
    bool owned = true;   // false: a logical duplicate of a device shares its twin's module and must not unload it
    unsigned far_waves = 0;  // waves per SIMD the unit's FAR pass is built for (0: the generic default)
    unsigned near_waves = 0, f32_waves = 0;   // … its NEAR / FULL passes, Float64 and Float32 (0: the generic defaults)
    hipFunction_t far = nullptr, near = nullptr, full10 = nullptr, fulln = nullptr, canvas = nullptr, prepare = nullptr,
                  eval_metric = nullptr, eval_geodesic = nullptr, eval_accel = nullptr;
    // Float32 twins (absent in units built without them)
    hipFunction_t full10_f32 = nullptr, fulln_f32 = nullptr, prepare_f32 = nullptr, canvas_f32 = nullptr;
    hipFunction_t redshift = nullptr, redshift_f32 = nullptr;   // optional (units built before round 3 have none)
    hipFunction_t resolve = nullptr, resolve_f32 = nullptr;     // the resolve kernel with the unit's objects (used when has_objects)
    hipFunction_t eval_objects = nullptr, eval_objects_f32 = nullptr;   // rtgr_eval_objects_* with the unit's objects
    hipFunction_t samples = nullptr;   // optional: one sample object per type the source offers one for (rtgr_user_sample)
    // what the unit was built for (rtgr_user_unit_desc): a scene runs with it only if its own variant is this one
    uint32_t metric = RTGR_USER;   // rtgr_metric (| RTGR_METRIC_GENERIC) of its kernels; RTGR_USER: a metric of its own
    bool spin = true;              // closed-form built-in kernels: the a != 0 instantiation
    bool has_metric = true, has_objects = false, has_reach = false;
    bool probe_ok = false;         // the load-time probe ran and passed (rtgr_units.hip: probe_unit)
};

struct TimedLaunch { hipEvent_t a, b; int which; };

// A scene whose list is longer than the kernels' argument block holds (DevScene::more): the whole list, its groups and runs of groups in
// one immutable device table per distinct (scalar type, list, layout) a device has seen, found again by content.  Immutable, so a
// kernel in flight — or a hipGraph captured earlier — never sees it change; freed by rtgr_trim / rtgr_destroy (or all at once, behind
// a device synchronisation, when a caller has gone through OBJECT_TABLES_MAX distinct lists or OBJECT_TABLES_BYTES of them: an
// animation of a 100000-object list is 9 MB a frame, on the device and in the host copy the lookup compares with).
struct ObjectTable { std::vector<char> content; void* dev = nullptr; };
constexpr size_t OBJECT_TABLES_MAX = 512;
constexpr size_t OBJECT_TABLES_BYTES = (size_t)1 << 30;

// pipeline workspace of one (device, stream)
struct StreamState {
    void* ws = nullptr;
    size_t ws_bytes = 0;
    unsigned long long* queue = nullptr;  // 8 work-queue heads; stream order makes one slot per stream enough
    std::vector<void*> retired;           // superseded workspaces: kept alive for graphs captured earlier
};

struct Staging;  // host entry points (rtgr_internal.hpp)

struct DeviceCtx {
    int dev = -1;       // HIP device ordinal
    int num_cu = 0;
    std::string name;
    std::mutex mu;      // held while a call enqueues its kernels: the enqueue sequences of two host threads never interleave
    std::unordered_map<hipStream_t, StreamState> streams;
    std::vector<UserModule> modules;
    std::unordered_multimap<uint64_t, ObjectTable> object_tables;   // by FNV-1a of the content
    size_t object_table_bytes = 0;                                   // … and what they hold together
    std::unordered_map<uint64_t, int> checked_scenes;                // auto_scene_check: key of a scene -> the verdict it got (RTGR_OK or the refusal)
    Knobs knobs;
    // optional per-kernel timing (bench.py's roofline leg): hipEvents around each kernel of the pipeline
    bool timing = false;
    std::vector<TimedLaunch> timed;
    double acc_ms[6] = {0, 0, 0, 0, 0, 0};       // summed durations since the last read: [0..3] rtgr_timing_read, [4..5] rtgr_timing_read_exchange
    uint64_t acc_n[6] = {0, 0, 0, 0, 0, 0};
    std::vector<hipEvent_t> event_pool;
    // the host-pointer entry points' pipelines: [0] every blocking call's; [1] the second frame in flight of rtgr_trace_frames_*
    std::unique_ptr<Staging, void (*)(Staging*)> staging{nullptr, nullptr}, staging2{nullptr, nullptr};
#ifdef RTGR_ROOT_STATS
    unsigned long long* dbg = nullptr;
#endif
    hipEvent_t take_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    const UserModule* find_module(uint64_t id) const {
        for (auto& m : modules) if (m.id == id) return &m;
        return nullptr;
    }
};

struct KernelTimer {  // RAII: records start now, stop at scope exit
    DeviceCtx& d; hipStream_t st; int which; hipEvent_t a = nullptr, b = nullptr;
    KernelTimer(DeviceCtx& dc, hipStream_t s, int w) : d(dc), st(s), which(w) {
        if (d.timing) { a = d.take_event(); b = d.take_event(); (void)hipEventRecord(a, st); }
    }
    ~KernelTimer() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        d.timed.push_back({a, b, which});
        // timing left on and never read: keep the newest launches only (their events go back to the pool)
        constexpr size_t CAP = 4096;
        if (d.timed.size() > CAP) {
            for (size_t i = 0; i < CAP / 2; i++) { d.event_pool.push_back(d.timed[i].a); d.event_pool.push_back(d.timed[i].b); }
            d.timed.erase(d.timed.begin(), d.timed.begin() + CAP / 2);
        }
    }
};

// what one pipeline launch needs from the context
struct LaunchEnv {
    DeviceCtx& d;
    StreamState& ss;
    const UserModule* user;  // the scene's unit (rtgr_scene.user_metric: RTGR_USER metric and / or RTGR_USER_OBJECT objects) or null
    hipEvent_t after_setup = nullptr;  // optional: recorded on the launch stream behind the ray set-up / queue-order kernels
    const Knobs* knobs = nullptr;      // this call's own launch options (the load-time probe, rtgr_scene_check); null: the device's
};

size_t align256(size_t b);
template <class R> size_t workspace_bytes(uint64_t rays, bool with_state);
template <class R> uint64_t pick_chunk(const DeviceCtx& d, const StreamState& ss, uint64_t n, bool with_state);
int ensure_workspace(DeviceCtx& d, StreamState& ss, size_t bytes, hipStream_t st);

// hipModuleLaunchKernel with the arguments given as C++ values
template <class... Args>
static inline hipError_t launch_module(hipFunction_t f, unsigned grid, unsigned block, hipStream_t st, Args... args) {
    void* params[] = {(void*)&args...};
    return hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, st, params, nullptr);
}

// ---- the pipeline, one function per translation unit (each TU instantiates the kernels of its metric variants) ----------
int launch_f64_mink(LaunchEnv& E, const TraceArgs<double>& A, hipStream_t st);
int launch_f64_ksref(LaunchEnv& E, const TraceArgs<double>& A, bool spin, hipStream_t st);
int launch_f64_kstrue(LaunchEnv& E, const TraceArgs<double>& A, bool spin, hipStream_t st);
int launch_f64_generic(LaunchEnv& E, const TraceArgs<double>& A, hipStream_t st);  // KS_REF / KS_TRUE / RTGR_USER
int launch_f32_closed(LaunchEnv& E, const TraceArgs<float>& A, bool spin, hipStream_t st);  // all three built-ins
int launch_f32_generic(LaunchEnv& E, const TraceArgs<float>& A, hipStream_t st);

// ---- small kernels (rtgr_misc.hip) -------------------------------------------------------------------------------------
int misc_canvas_f64(const DevScene<double>& sc, const DevCamera<double>& cam, uint64_t ni, uint64_t nj, uint64_t j0,
                    uint64_t n, double* d_state0, hipStream_t st);
int misc_canvas_f32(const DevScene<float>& sc, const DevCamera<float>& cam, uint64_t ni, uint64_t nj, uint64_t j0,
                    uint64_t n, float* d_state0, hipStream_t st);
int misc_eval_metric_f64(const DevScene<double>& sc, const double* d_x, uint64_t n, double* g, double* dg, double* Gam, hipStream_t st);
int misc_eval_metric_f32(const DevScene<float>& sc, const float* d_x, uint64_t n, float* g, float* dg, float* Gam, hipStream_t st);
int misc_eval_geodesic_f64(const DevScene<double>& sc, const double* d_s, uint64_t n, int path, double* d_ds, hipStream_t st);
int misc_eval_geodesic_f32(const DevScene<float>& sc, const float* d_s, uint64_t n, int path, float* d_ds, hipStream_t st);
int misc_eval_objects_f64(const DevScene<double>& sc, const DevSolver<double>& opt, const double* d_x, uint64_t n, double* d, double* dmin, uint8_t* hit,
                          double* rgb, hipStream_t st);
int misc_eval_objects_f32(const DevScene<float>& sc, const DevSolver<float>& opt, const float* d_x, uint64_t n, float* d, float* dmin, uint8_t* hit,
                          float* rgb, hipStream_t st);
int misc_eval_fastmath_f64(const double* d_x, uint64_t n, double* d_rcp, double* d_rsq, hipStream_t st);
int misc_redshift_f64(const DevScene<double>& sc, const DevCamera<double>& cam, const double* d_state0, uint64_t ni, uint64_t nj,
                      uint64_t j0, uint64_t jstride, uint64_t n, uint64_t out_offset, const double* d_state_end,
                      const uint8_t* d_hit, const uint32_t* d_hit32, double* d_red, hipStream_t st);
int misc_redshift_f32(const DevScene<float>& sc, const DevCamera<float>& cam, const float* d_state0, uint64_t ni, uint64_t nj,
                      uint64_t j0, uint64_t jstride, uint64_t n, uint64_t out_offset, const float* d_state_end,
                      const uint8_t* d_hit, const uint32_t* d_hit32, float* d_red, hipStream_t st);
int misc_quantize(const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, hipStream_t st);
// multi-device gather on device 0: rows of rank r (cyclic over nranks) back into place
int misc_place_rows_f64(const double* d_part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks, uint64_t planes,
                        double* d_full, hipStream_t st);
int misc_place_rows_f32(const float* d_part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks, uint64_t planes,
                        float* d_full, hipStream_t st);
int misc_place_rows_u8(const uint8_t* d_part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks, uint64_t elem,
                       uint8_t* d_full, hipStream_t st);
int misc_poison_registers(int n_cu, unsigned pattern, hipStream_t st);   // the load-time probe's scrubber (rtgr_misc.hip)

}  // namespace rtgr
