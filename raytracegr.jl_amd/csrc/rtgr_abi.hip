// rtgr_abi.hip — the C ABI of include/rtgr.h: every exported symbol, and nothing else.  Each is a one-line shim onto the function of
// the same name (minus the rtgr_ prefix) in namespace rtgr::api, defined in rtgr_context.hip (lifecycle, options, timing, device entry
// points), rtgr_host_pipeline.hip (host-pointer entry points), rtgr_sharded.hip (all devices of a context), rtgr_hooks.hip (camera,
// parity hooks, quantisation) and rtgr_units.hip (run-time units).  tests/test_abi.py holds this list against include/rtgr.h.
#include "rtgr_internal.hpp"

extern "C" {

int rtgr_create(const int* device_ids, int n_devices, rtgr_context** ctx_out) { return rtgr::api::create(device_ids, n_devices, ctx_out); }
int rtgr_destroy(rtgr_context* ctx) { return rtgr::api::destroy(ctx); }
int rtgr_context_devices(rtgr_context* ctx) { return rtgr::api::context_devices(ctx); }
int rtgr_trim(rtgr_context* ctx) { return rtgr::api::trim(ctx); }
int rtgr_init(int device) { return rtgr::api::init(device); }
int rtgr_shutdown(void) { return rtgr::api::shutdown(); }
const char* rtgr_last_error(void) { return rtgr::api::last_error(); }
int rtgr_abi_version(void) { return rtgr::api::abi_version(); }
int rtgr_solver_defaults(rtgr_solver* s, int is_f32) { return rtgr::api::solver_defaults(s, is_f32); }
int rtgr_device_info(rtgr_context* ctx, int index, char* name, uint64_t name_len, int* n_cu, int* clock_mhz, int* wavefront) { return rtgr::api::device_info(ctx, index, name, name_len, n_cu, clock_mhz, wavefront); }
int rtgr_set_option(rtgr_context* ctx, const char* name, long value) { return rtgr::api::set_option(ctx, name, value); }
int rtgr_get_option(rtgr_context* ctx, const char* name, long* value) { return rtgr::api::get_option(ctx, name, value); }
int rtgr_reserve_workspace(rtgr_context* ctx, const void* d_any, void* stream, uint64_t n_rays, int with_state_end, int is_f32) { return rtgr::api::reserve_workspace(ctx, d_any, stream, n_rays, with_state_end, is_f32); }
int rtgr_timing_enable(rtgr_context* ctx, int index, int on) { return rtgr::api::timing_enable(ctx, index, on); }
int rtgr_timing_read(rtgr_context* ctx, int index, double ms[4], uint64_t launches[4]) { return rtgr::api::timing_read(ctx, index, ms, launches); }
int rtgr_timing_read_exchange(rtgr_context* ctx, int index, double ms[2], uint64_t launches[2]) { return rtgr::api::timing_read_exchange(ctx, index, ms, launches); }
int rtgr_peer_access(rtgr_context* ctx, int index, char* why, uint64_t why_len) { return rtgr::api::peer_access(ctx, index, why, why_len); }
int rtgr_trace_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* d_state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) { return rtgr::api::trace_device_f64(ctx, scene, opt, d_state0, cam, ni, nj, j0, j1, d_rgb, out, d_counters, stream); }
int rtgr_trace_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* d_state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) { return rtgr::api::trace_device_f32(ctx, scene, opt, d_state0, cam, ni, nj, j0, j1, d_rgb, out, d_counters, stream); }
int rtgr_trace_rows_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) { return rtgr::api::trace_rows_device_f64(ctx, scene, opt, cam, ni, nj, j0, jstride, nrows, d_rgb, out, d_counters, stream); }
int rtgr_trace_rows_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) { return rtgr::api::trace_rows_device_f32(ctx, scene, opt, cam, ni, nj, j0, jstride, nrows, d_rgb, out, d_counters, stream); }
int rtgr_trace_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) { return rtgr::api::trace_f64(ctx, scene, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr); }
int rtgr_trace_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* state0, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) { return rtgr::api::trace_f32(ctx, scene, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr); }
int rtgr_trace_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* pixels_in, uint64_t ni, uint64_t nj, double* pixels_out, rtgr_counters* ctr) { return rtgr::api::trace_pixels_f64(ctx, scene, opt, pixels_in, ni, nj, pixels_out, ctr); }
int rtgr_trace_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* pixels_in, uint64_t ni, uint64_t nj, float* pixels_out, rtgr_counters* ctr) { return rtgr::api::trace_pixels_f32(ctx, scene, opt, pixels_in, ni, nj, pixels_out, ctr); }
int rtgr_trace_frames_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams, const double* const* state0s, uint64_t ni, uint64_t nj, double* const* rgb, const rtgr_ray_outputs* outs, rtgr_counters* ctrs) { return rtgr::api::trace_frames_f64(ctx, scene, opt, nframes, cams, state0s, ni, nj, rgb, outs, ctrs); }
int rtgr_trace_frames_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams, const float* const* state0s, uint64_t ni, uint64_t nj, float* const* rgb, const rtgr_ray_outputs* outs, rtgr_counters* ctrs) { return rtgr::api::trace_frames_f32(ctx, scene, opt, nframes, cams, state0s, ni, nj, rgb, outs, ctrs); }
int rtgr_trace_frames_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const double* const* pixels_in, uint64_t ni, uint64_t nj, double* const* pixels_out, rtgr_counters* ctrs) { return rtgr::api::trace_frames_pixels_f64(ctx, scene, opt, nframes, pixels_in, ni, nj, pixels_out, ctrs); }
int rtgr_trace_frames_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const float* const* pixels_in, uint64_t ni, uint64_t nj, float* const* pixels_out, rtgr_counters* ctrs) { return rtgr::api::trace_frames_pixels_f32(ctx, scene, opt, nframes, pixels_in, ni, nj, pixels_out, ctrs); }
int rtgr_trace_one_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double pos[4], const double normal[4], double rgb[3], double state_end[8], uint8_t* status) { return rtgr::api::trace_one_f64(ctx, scene, opt, pos, normal, rgb, state_end, status); }
int rtgr_trace_one_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float pos[4], const float normal[4], float rgb[3], float state_end[8], uint8_t* status) { return rtgr::api::trace_one_f32(ctx, scene, opt, pos, normal, rgb, state_end, status); }
int rtgr_trace_sharded_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, double* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) { return rtgr::api::trace_sharded_f64(ctx, scene, opt, cam, ni, nj, rgb, out, ctr); }
int rtgr_trace_sharded_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) { return rtgr::api::trace_sharded_device_f64(ctx, scene, opt, cam, ni, nj, d_rgb, out, ctr); }
int rtgr_trace_sharded_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, float* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) { return rtgr::api::trace_sharded_f32(ctx, scene, opt, cam, ni, nj, rgb, out, ctr); }
int rtgr_trace_sharded_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) { return rtgr::api::trace_sharded_device_f32(ctx, scene, opt, cam, ni, nj, d_rgb, out, ctr); }
int rtgr_make_canvas_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* d_state0, void* stream) { return rtgr::api::make_canvas_device_f64(ctx, scene, cam, ni, nj, j0, j1, d_state0, stream); }
int rtgr_make_canvas_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* state0) { return rtgr::api::make_canvas_f64(ctx, scene, cam, ni, nj, j0, j1, state0); }
int rtgr_make_canvas_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* d_state0, void* stream) { return rtgr::api::make_canvas_device_f32(ctx, scene, cam, ni, nj, j0, j1, d_state0, stream); }
int rtgr_make_canvas_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* state0) { return rtgr::api::make_canvas_f32(ctx, scene, cam, ni, nj, j0, j1, state0); }
int rtgr_eval_metric_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* x , uint64_t n, double* g, double* dg, double* Gam) { return rtgr::api::eval_metric_f64(ctx, scene, x, n, g, dg, Gam); }
int rtgr_eval_metric_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* x , uint64_t n, float* g, float* dg, float* Gam) { return rtgr::api::eval_metric_f32(ctx, scene, x, n, g, dg, Gam); }
int rtgr_eval_geodesic_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* s , uint64_t n, int path, double* ds) { return rtgr::api::eval_geodesic_f64(ctx, scene, s, n, path, ds); }
int rtgr_eval_geodesic_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* s , uint64_t n, int path, float* ds) { return rtgr::api::eval_geodesic_f32(ctx, scene, s, n, path, ds); }
int rtgr_eval_objects_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* x , uint64_t n, double* d, double* dmin, uint8_t* hit, double* rgb) { return rtgr::api::eval_objects_f64(ctx, scene, opt, x, n, d, dmin, hit, rgb); }
int rtgr_eval_objects_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* x , uint64_t n, float* d, float* dmin, uint8_t* hit, float* rgb) { return rtgr::api::eval_objects_f32(ctx, scene, opt, x, n, d, dmin, hit, rgb); }
int rtgr_eval_fastmath_f64(rtgr_context* ctx, const double* x, uint64_t n, double* rcp, double* rsq) { return rtgr::api::eval_fastmath_f64(ctx, x, n, rcp, rsq); }
int rtgr_user_metric_load(rtgr_context* ctx, const char* code_object_path, uint64_t* id_out) { return rtgr::api::user_metric_load(ctx, code_object_path, id_out); }
int rtgr_user_metric_compile(rtgr_context* ctx, const char* source, int stationary, uint64_t* id_out) { return rtgr::api::user_metric_compile(ctx, source, stationary, id_out); }
int rtgr_user_metric_build(const char* source, int stationary, const char* code_object_path) { return rtgr::api::user_metric_build(source, stationary, code_object_path); }
int rtgr_user_metric_unload(rtgr_context* ctx, uint64_t id) { return rtgr::api::user_metric_unload(ctx, id); }
int rtgr_user_unit_compile(rtgr_context* ctx, const char* source, int stationary, const rtgr_scene* built_for, uint64_t* id_out) { return rtgr::api::user_unit_compile(ctx, source, stationary, built_for, id_out); }
int rtgr_user_unit_build(const char* source, int stationary, const rtgr_scene* built_for, const char* code_object_path) { return rtgr::api::user_unit_build(source, stationary, built_for, code_object_path); }
int rtgr_user_source_join(const char* const* sources, const uint32_t* ntypes, int n, char* out, uint64_t cap, uint64_t* need) { return rtgr::api::user_source_join(sources, ntypes, n, out, cap, need); }
int rtgr_user_unit_info(rtgr_context* ctx, uint64_t id, rtgr_unit_info* info) { return rtgr::api::user_unit_info(ctx, id, info); }
int rtgr_scene_check(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, int is_f32) { return rtgr::api::scene_check(ctx, scene, opt, cam, ni, nj, is_f32); }
int rtgr_code_object_audit(const char* code_object_path, int* found, char* report, uint64_t report_len) { return rtgr::api::code_object_audit(code_object_path, found, report, report_len); }
int rtgr_listing_repair(const char* listing_path, const char* repaired_path, int* blocks) { return rtgr::api::listing_repair(listing_path, repaired_path, blocks); }
int rtgr_user_metric_loaded(rtgr_context* ctx, uint64_t id) { return rtgr::api::user_metric_loaded(ctx, id); }
int rtgr_quantize_device_f64(rtgr_context* ctx, const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, void* stream) { return rtgr::api::quantize_device_f64(ctx, d_rgb, ni, nj, d_img, stream); }

}  // extern "C"
