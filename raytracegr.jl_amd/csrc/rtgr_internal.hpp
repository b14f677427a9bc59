// rtgr_internal.hpp — what the HOST translation units of librtgr_hip.so share (no kernel, no device code):
//
//   rtgr_context.hip        error plumbing, launch options, contexts and their per-device / per-stream state, argument checking
//                           and conversion, dispatch to the kernels' translation units (tu_*.hip), the device entry points
//   rtgr_host_pipeline.hip  the host-pointer entry points: pinned staging + a three-stream H2D / compute / D2H pipeline
//   rtgr_sharded.hip        one call over every device of a context: cyclic rows, peer copies to device 0
//   rtgr_hooks.hip          make_canvas and the parity hooks (rtgr_eval_*), image quantisation
//   rtgr_units.hip          run-time compiled units: build, audit, load-time probe, load, scene check
//   rtgr_abi.hip            the `extern "C"` symbols of include/rtgr.h, each a one-line shim onto rtgr::api::<name>
//
// Everything here is internal to the library: nothing in this header crosses the C ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <thread>

#include "rtgr_host.hpp"

namespace rtgr {

// ---- staging of the host-pointer entry points and of the multi-device path (one per DeviceCtx, created on first use) ----------
struct PinnedBuf {
    void* p = nullptr; size_t bytes = 0;
    int need(size_t b) {
        if (b <= bytes) return RTGR_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        HIP_TRY(hipHostMalloc(&p, b, hipHostMallocDefault));
        bytes = b;
        return RTGR_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};
struct DevBufG {  // grow-only device buffer
    void* p = nullptr; size_t bytes = 0;
    int need(size_t b) {
        if (b <= bytes) return RTGR_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }  // only called with the owning streams idle
        HIP_TRY(hipMalloc(&p, b));
        bytes = b;
        return RTGR_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// staging of the host-pointer entry points and of the multi-device path (one per DeviceCtx, created on first use)
struct Staging {
    hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
    static constexpr int IN_SLOTS = 3, OUT_SLOTS = 2;
    PinnedBuf pin_in[IN_SLOTS], pin_out[OUT_SLOTS], pin_small;
    hipEvent_t ev_in[IN_SLOTS] = {nullptr, nullptr, nullptr};
    DevBufG d_in, d_out, d_small;  // device-side inputs (ray states), outputs (all requested arrays), counters + flags
    DevBufG d_recv;                // device 0 of a multi-device context: rows received from the peers
    std::mutex mu;                 // one host-pointer call at a time per device (they share the staging buffers)
};
void staging_delete(Staging* s);

}  // namespace rtgr

struct rtgr_context {
    std::vector<std::unique_ptr<rtgr::DeviceCtx>> devs;
    // peer access device 0 <-> device k, established by rtgr_create: 1 = enabled both ways (or the same physical device),
    // 0 = not available, with the reason (the multi-device gather then stages through pinned host memory, or fails when the
    // option peer = 1 requires peer copies)
    std::vector<char> peer_ok;
    std::vector<std::string> peer_why;
    std::mutex modules_mu;   // run-time metric modules are loaded / unloaded on all devices under ONE lock
    // rtgr_user_unit_compile: hash of (source, what it is built for, device headers) -> id of the unit it gave; the same call again is
    // answered from here while that unit is resident (a C or Julia caller need not keep a table of its own to avoid a 5 s rebuild)
    std::mutex compiled_mu;
    std::unordered_map<uint64_t, uint64_t> compiled;
};

namespace rtgr {

struct DeviceGuard {  // the calling thread's current device is restored on scope exit
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- contexts (rtgr_context.hip) -------------------------------------------------------------------------------------------
extern std::mutex g_default_mu;      // guards g_default
extern rtgr_context* g_default;      // the process's default context (ctx == NULL), created on first use
const std::string& last_error_string();   // the calling thread's last failure message
int staging_of(DeviceCtx& d, Staging** out, int slot = 0);   // slot 1: the second frame in flight (rtgr_trace_frames_*)
void free_device_state(DeviceCtx& d, bool all);
int create_context(const int* ids, int n, rtgr_context** out);
void destroy_context(rtgr_context* c);
int resolve_ctx(rtgr_context* in, rtgr_context** out);   // ctx == NULL: the default context
int device_of(rtgr_context* c, const void* d_ptr, DeviceCtx** out);
int stream_state(DeviceCtx& d, hipStream_t st, StreamState** out);
int collect_timed(DeviceCtx& d);
uint64_t fnv1a(const std::vector<char>& b);

#define RESOLVE_DEVICE(ptr)                          \
    rtgr_context* c = nullptr;                       \
    int rc = resolve_ctx(ctx, &c);                   \
    if (rc) return rc;                               \
    DeviceCtx* D = nullptr;                          \
    if ((rc = device_of(c, (ptr), &D))) return rc

// ---- argument conversion and the device-side enqueue (rtgr_context.hip) ------------------------------------------------------
void scene_variant(const rtgr_scene* s, uint32_t* metric, bool* spin);
// The load-time probe of a unit of OBJECTS traces a scene of built-in objects (it knows no parameters of the user's) and must still run
// the UNIT's kernels, not the library's: while this is set on the calling thread, a scene that names a unit runs with it even though
// nothing in the scene requires one.  (Everywhere else a built-in scene ignores rtgr_scene.user_metric, as it always has.)
extern thread_local bool tl_probe_forces_unit;
// … and the probe (and rtgr_scene_check) choose the launch options of THEIR calls — pass structure, queue order, grid size — without
// touching the device's options, which other threads' calls on the same device read: null = the device's options decide.
extern thread_local const Knobs* tl_knobs_override;
// (D.mu held.  `st`: the stream the scene's kernels will be enqueued on — a list of more than RTGR_MAX_OBJECTS objects seen for the
//  first time is uploaded to a device table, which a stream under capture cannot wait for)
template <class R> int convert_scene(DeviceCtx& D, const rtgr_scene* s, DevScene<R>& d, const UserModule** user, hipStream_t st = nullptr);
// the object list of a scene: rtgr_scene.objects when given, else the inline slots
inline const rtgr_object* scene_objects(const rtgr_scene* s) { return s->objects ? s->objects : s->obj; }
template <class R> int convert_solver(const rtgr_solver* s, DevSolver<R>& d);
template <class R> void convert_camera(const rtgr_camera* c, DevCamera<R>& d);
// window of a larger output: see TraceArgs::plane_stride / out_offset
struct Window { uint64_t plane_stride = 0, out_offset = 0; uint32_t* nan_flag = nullptr; hipEvent_t after_setup = nullptr; };
// Enqueue the pipeline for rows of a canvas on device D, stream st.  The caller holds no lock; this takes D.mu for the
// duration of the enqueue.
template <class R>
int trace_device(DeviceCtx& D, const rtgr_scene* scene, const rtgr_solver* opt, const R* d_state0, const rtgr_camera* cam,
                 uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* d_rgb, const rtgr_ray_outputs* out,
                 rtgr_counters* d_counters, hipStream_t st, uint64_t jstride = 1, uint64_t nrows_strided = 0,
                 const Window* win = nullptr);

// ---- the host-pointer path (rtgr_host_pipeline.hip) ---------------------------------------------------------------------------
template <class R>
int trace_host(rtgr_context* ctx_in, const rtgr_scene* scene, const rtgr_solver* opt, const R* state0, const rtgr_camera* cam,
               uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);

// scratch device buffers of the small host-pointer hooks (eval_*, make_canvas, the probe): RAII, synchronous
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
int host_has_nan(const R* v, uint64_t count) {
    for (uint64_t q = 0; q < count; q++)
        if (v[q] != v[q]) return 1;
    return 0;
}

// ---- run-time units (rtgr_units.hip) ---------------------------------------------------------------------------------------------
int unload_locked(rtgr_context* c, uint64_t id);   // drop unit `id` (0: all) from every device; the caller holds c->modules_mu
// The scene check nobody has to remember: called by trace_device ahead of every enqueue, answers from the device's
// table of checked scenes in all but the first call for a scene.  D.mu NOT held.
template <class R>
int auto_scene_check(DeviceCtx& D, const rtgr_scene* scene, const rtgr_solver* opt, const R* d_state0, const rtgr_camera* cam,
                     uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, uint64_t jstride, uint64_t nrows_strided, hipStream_t st);

// ---- the entry points of include/rtgr.h, by the name behind the rtgr_ prefix (rtgr_abi.hip holds the extern "C" shims) ----------
namespace api {
int create(const int* device_ids, int n_devices, rtgr_context** ctx_out);
int destroy(rtgr_context* ctx);
int context_devices(rtgr_context* ctx);
int trim(rtgr_context* ctx);
int init(int device);
int shutdown(void);
const char* last_error(void);
int abi_version(void);
int solver_defaults(rtgr_solver* s, int is_f32);
int device_info(rtgr_context* ctx, int index, char* name, uint64_t name_len, int* n_cu, int* clock_mhz, int* wavefront);
int set_option(rtgr_context* ctx, const char* name, long value);
int get_option(rtgr_context* ctx, const char* name, long* value);
int reserve_workspace(rtgr_context* ctx, const void* d_any, void* stream, uint64_t n_rays, int with_state_end, int is_f32);
int timing_enable(rtgr_context* ctx, int index, int on);
int timing_read(rtgr_context* ctx, int index, double ms[4], uint64_t launches[4]);
int timing_read_exchange(rtgr_context* ctx, int index, double ms[2], uint64_t launches[2]);
int peer_access(rtgr_context* ctx, int index, char* why, uint64_t why_len);
int trace_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* d_state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);
int trace_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* d_state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);
int trace_rows_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);
int trace_rows_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);
int trace_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int trace_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int trace_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* pixels_in, uint64_t ni, uint64_t nj, double* pixels_out, rtgr_counters* ctr);
int trace_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* pixels_in, uint64_t ni, uint64_t nj, float* pixels_out, rtgr_counters* ctr);
int trace_frames_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams, const double* const* state0s, uint64_t ni, uint64_t nj, double* const* rgb, const rtgr_ray_outputs* outs, rtgr_counters* ctrs);
int trace_frames_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams, const float* const* state0s, uint64_t ni, uint64_t nj, float* const* rgb, const rtgr_ray_outputs* outs, rtgr_counters* ctrs);
int trace_frames_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const double* const* pixels_in, uint64_t ni, uint64_t nj, double* const* pixels_out, rtgr_counters* ctrs);
int trace_frames_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const float* const* pixels_in, uint64_t ni, uint64_t nj, float* const* pixels_out, rtgr_counters* ctrs);
int trace_one_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double pos[4], const double normal[4], double rgb[3], double state_end[8], uint8_t* status);
int trace_one_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float pos[4], const float normal[4], float rgb[3], float state_end[8], uint8_t* status);
int trace_sharded_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, double* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int trace_sharded_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int trace_sharded_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, float* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int trace_sharded_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int make_canvas_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* d_state0, void* stream);
int make_canvas_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* state0);
int make_canvas_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* d_state0, void* stream);
int make_canvas_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* state0);
int eval_metric_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* x , uint64_t n, double* g, double* dg, double* Gam);
int eval_metric_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* x , uint64_t n, float* g, float* dg, float* Gam);
int eval_geodesic_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* s , uint64_t n, int path, double* ds);
int eval_geodesic_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* s , uint64_t n, int path, float* ds);
int eval_objects_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* x , uint64_t n, double* d, double* dmin, uint8_t* hit, double* rgb);
int eval_objects_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* x , uint64_t n, float* d, float* dmin, uint8_t* hit, float* rgb);
int eval_fastmath_f64(rtgr_context* ctx, const double* x, uint64_t n, double* rcp, double* rsq);
int user_metric_load(rtgr_context* ctx, const char* code_object_path, uint64_t* id_out);
int user_metric_compile(rtgr_context* ctx, const char* source, int stationary, uint64_t* id_out);
int user_metric_build(const char* source, int stationary, const char* code_object_path);
int user_metric_unload(rtgr_context* ctx, uint64_t id);
int user_unit_compile(rtgr_context* ctx, const char* source, int stationary, const rtgr_scene* built_for, uint64_t* id_out);
int user_unit_build(const char* source, int stationary, const rtgr_scene* built_for, const char* code_object_path);
int user_source_join(const char* const* sources, const uint32_t* ntypes, int n, char* out, uint64_t cap, uint64_t* need);
int user_unit_info(rtgr_context* ctx, uint64_t id, rtgr_unit_info* info);
int scene_check(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, int is_f32);
int code_object_audit(const char* code_object_path, int* found, char* report, uint64_t report_len);
int listing_repair(const char* listing_path, const char* repaired_path, int* blocks);
int user_metric_loaded(rtgr_context* ctx, uint64_t id);
int quantize_device_f64(rtgr_context* ctx, const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, void* stream);
}  // namespace api

}  // namespace rtgr
