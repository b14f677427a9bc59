// rtgr_hooks.hip — the small entry points around the hot path: make_canvas (src/RayTraceGR.jl:457-478), the parity hooks the reference's
// unit tests exercise (metric / dmetric / christoffel / geodesic, test/runtests.jl:12-61; objects and the colour rule), N0f8
// quantisation.  Host side only: the kernels are rtgr_misc.hip's, or the scene's unit's.
#include "rtgr_internal.hpp"

namespace rtgr {

// ---- camera / hooks ------------------------------------------------------------------------------------------------------
template <class R>
static int make_canvas_device(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj,
                              uint64_t j0, uint64_t j1, R* d_state0, void* stream) {
    if (!cam || !d_state0) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    RESOLVE_DEVICE(d_state0);
    if (ni == 0 || nj == 0 || j1 <= j0 || j1 > nj) return fail(RTGR_ERR_BAD_ARG, "bad canvas range");
    DeviceGuard guard(D->dev);
    std::lock_guard<std::mutex> lk(D->mu);
    DevScene<R> sc;
    const UserModule* user = nullptr;
    if ((rc = convert_scene<R>(*D, scene, sc, &user))) return rc;
    DevCamera<R> cm;
    convert_camera<R>(cam, cm);
    const uint64_t n = ni * (j1 - j0);
    if (sc.metric == RTGR_USER) {
        hipFunction_t f = sizeof(R) == 8 ? user->canvas : user->canvas_f32;
        if (!f) return fail(RTGR_ERR_BAD_ARG, "this user-metric code object carries no Float32 kernels");
        HIP_TRY(launch_module(f, (unsigned)((n + 255) / 256), 256, (hipStream_t)stream, sc, cm, ni, nj, j0, (uint64_t)1,
                              (uint64_t)0, n, d_state0));
        return RTGR_OK;
    }
    if constexpr (sizeof(R) == 8) return misc_canvas_f64(sc, cm, ni, nj, j0, n, d_state0, (hipStream_t)stream);
    else return misc_canvas_f32(sc, cm, ni, nj, j0, n, d_state0, (hipStream_t)stream);
}
template <class R>
static int make_canvas_host(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                            uint64_t j1, R* state0) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!state0) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (ni == 0 || nj == 0 || j1 <= j0 || j1 > nj) return fail(RTGR_ERR_BAD_ARG, "bad canvas range");
    const uint64_t n = ni * (j1 - j0);
    DeviceGuard guard(c->devs[0]->dev);
    DevBuf b;
    if ((rc = b.alloc(n * 8 * sizeof(R)))) return rc;
    if ((rc = make_canvas_device<R>(c, scene, cam, ni, nj, j0, j1, (R*)b.p, nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(state0, b.p, n * 8 * sizeof(R), hipMemcpyDeviceToHost));
    return RTGR_OK;
}
int api::make_canvas_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj,
                                uint64_t j0, uint64_t j1, double* d_state0, void* stream) {
    return make_canvas_device<double>(ctx, scene, cam, ni, nj, j0, j1, d_state0, stream);
}
int api::make_canvas_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj,
                                uint64_t j0, uint64_t j1, float* d_state0, void* stream) {
    return make_canvas_device<float>(ctx, scene, cam, ni, nj, j0, j1, d_state0, stream);
}
int api::make_canvas_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                         uint64_t j1, double* state0) {
    return make_canvas_host<double>(ctx, scene, cam, ni, nj, j0, j1, state0);
}
int api::make_canvas_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0,
                         uint64_t j1, float* state0) {
    return make_canvas_host<float>(ctx, scene, cam, ni, nj, j0, j1, state0);
}

template <class R>
static int eval_metric_host(rtgr_context* ctx, const rtgr_scene* scene, const R* x, uint64_t n, R* g, R* dg, R* Gam) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!x) return fail(RTGR_ERR_BAD_ARG, "x is NULL");
    if (n == 0) return RTGR_OK;
    if (host_has_nan(x, 4 * n)) return fail(RTGR_ERR_NAN_INPUT, "NaN coordinate (AssertionError in the reference, :279)");
    DeviceCtx& D = *c->devs[0];
    DeviceGuard guard(D.dev);
    std::lock_guard<std::mutex> lk(D.mu);
    DevScene<R> sc;
    const UserModule* user = nullptr;
    if ((rc = convert_scene<R>(D, scene, sc, &user))) return rc;
    DevBuf bx, bg, bd, bG;
    if ((rc = bx.alloc(n * 4 * sizeof(R)))) return rc;
    HIP_TRY(hipMemcpy(bx.p, x, n * 4 * sizeof(R), hipMemcpyHostToDevice));
    if (g && (rc = bg.alloc(n * 16 * sizeof(R)))) return rc;
    if (dg && (rc = bd.alloc(n * 64 * sizeof(R)))) return rc;
    if (Gam && (rc = bG.alloc(n * 64 * sizeof(R)))) return rc;
    if (sc.metric == RTGR_USER) {
        if (sizeof(R) != 8) return fail(RTGR_ERR_BAD_ARG, "user metrics are evaluated in Float64");
        HIP_TRY(launch_module(user->eval_metric, (unsigned)((n + 255) / 256), 256, (hipStream_t) nullptr, sc, (const R*)bx.p, n,
                              (R*)bg.p, (R*)bd.p, (R*)bG.p));
    } else if constexpr (sizeof(R) == 8) {
        if ((rc = misc_eval_metric_f64(sc, (const double*)bx.p, n, (double*)bg.p, (double*)bd.p, (double*)bG.p, nullptr))) return rc;
    } else {
        if ((rc = misc_eval_metric_f32(sc, (const float*)bx.p, n, (float*)bg.p, (float*)bd.p, (float*)bG.p, nullptr))) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (g) HIP_TRY(hipMemcpy(g, bg.p, n * 16 * sizeof(R), hipMemcpyDeviceToHost));
    if (dg) HIP_TRY(hipMemcpy(dg, bd.p, n * 64 * sizeof(R), hipMemcpyDeviceToHost));
    if (Gam) HIP_TRY(hipMemcpy(Gam, bG.p, n * 64 * sizeof(R), hipMemcpyDeviceToHost));
    return RTGR_OK;
}
int api::eval_metric_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* x, uint64_t n, double* g, double* dg, double* Gam) {
    return eval_metric_host<double>(ctx, scene, x, n, g, dg, Gam);
}
int api::eval_metric_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* x, uint64_t n, float* g, float* dg, float* Gam) {
    return eval_metric_host<float>(ctx, scene, x, n, g, dg, Gam);
}

template <class R>
static int eval_geodesic_host(rtgr_context* ctx, const rtgr_scene* scene, const R* s, uint64_t n, int path, R* ds) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!s || !ds) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (path < 0 || path > 2)
        return fail(RTGR_ERR_BAD_ARG, "path must be 0 (closed contraction, IEEE division), 1 (generic duals) or 2 (the integrate loop's own RHS)");
    if (n == 0) return RTGR_OK;
    if (host_has_nan(s, 8 * n)) return fail(RTGR_ERR_NAN_INPUT, "NaN state (AssertionError in the reference, :279)");
    DeviceCtx& D = *c->devs[0];
    DeviceGuard guard(D.dev);
    std::lock_guard<std::mutex> lk(D.mu);
    DevScene<R> sc;
    const UserModule* user = nullptr;
    if ((rc = convert_scene<R>(D, scene, sc, &user))) return rc;
    DevBuf bi, bo;
    if ((rc = bi.alloc(n * 8 * sizeof(R)))) return rc;
    if ((rc = bo.alloc(n * 8 * sizeof(R)))) return rc;
    HIP_TRY(hipMemcpy(bi.p, s, n * 8 * sizeof(R), hipMemcpyHostToDevice));
    if (sc.metric == RTGR_USER) {  // paths 0 / 1: the reference formulation (4-wide duals through g); path 2: the unit's own loop RHS
        if (sizeof(R) != 8) return fail(RTGR_ERR_BAD_ARG, "user metrics are evaluated in Float64");
        if (path == 2 && !user->eval_accel) return fail(RTGR_ERR_BAD_ARG, "this user-metric code object carries no rtgr_user_eval_accel");
        HIP_TRY(launch_module(path == 2 ? user->eval_accel : user->eval_geodesic, (unsigned)((n + 255) / 256), 256,
                              (hipStream_t) nullptr, sc, (const R*)bi.p, n, (R*)bo.p));
    } else if constexpr (sizeof(R) == 8) {
        if ((rc = misc_eval_geodesic_f64(sc, (const double*)bi.p, n, path, (double*)bo.p, nullptr))) return rc;
    } else {
        if ((rc = misc_eval_geodesic_f32(sc, (const float*)bi.p, n, path, (float*)bo.p, nullptr))) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ds, bo.p, n * 8 * sizeof(R), hipMemcpyDeviceToHost));
    return RTGR_OK;
}
template <class R>
static int eval_objects_host(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const R* x, uint64_t n, R* d, R* dmin, uint8_t* hit, R* rgb) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!x) return fail(RTGR_ERR_BAD_ARG, "x is NULL");
    if (n == 0) return RTGR_OK;
    DeviceCtx& D = *c->devs[0];
    DeviceGuard guard(D.dev);
    std::lock_guard<std::mutex> lk(D.mu);
    DevScene<R> sc;
    DevSolver<R> so;
    const UserModule* user = nullptr;
    if ((rc = convert_scene<R>(D, scene, sc, &user))) return rc;
    if ((rc = convert_solver<R>(opt, so))) return rc;
    if (hit && sc.nobj > 255u) return fail(RTGR_ERR_BAD_ARG, "rtgr_eval_objects: `hit` is a byte per point and the scene has more than 255 objects");
    const size_t nd = (size_t)n * (sc.nobj ? sc.nobj : 1);
    DevBuf bx, bd, bm, bh, bc;
    if ((rc = bx.alloc(n * 4 * sizeof(R)))) return rc;
    HIP_TRY(hipMemcpy(bx.p, x, n * 4 * sizeof(R), hipMemcpyHostToDevice));
    if (d && (rc = bd.alloc(nd * sizeof(R)))) return rc;
    if (dmin && (rc = bm.alloc(n * sizeof(R)))) return rc;
    if (hit && (rc = bh.alloc(n))) return rc;
    if (rgb && (rc = bc.alloc(n * 3 * sizeof(R)))) return rc;
    if (user && user->has_objects) {   // the scene's unit knows its objects' methods: its kernel
        hipFunction_t f = sizeof(R) == 8 ? user->eval_objects : user->eval_objects_f32;
        if (!f) return fail(RTGR_ERR_BAD_ARG, "this unit carries no rtgr_user_eval_objects kernel (rebuild the unit)");
        HIP_TRY(launch_module(f, (unsigned)((n + 255) / 256), 256, (hipStream_t) nullptr, sc, so, (const R*)bx.p, n, (R*)bd.p, (R*)bm.p, (uint8_t*)bh.p, (R*)bc.p));
    } else if constexpr (sizeof(R) == 8) {
        if ((rc = misc_eval_objects_f64(sc, so, (const double*)bx.p, n, (double*)bd.p, (double*)bm.p, (uint8_t*)bh.p, (double*)bc.p, nullptr))) return rc;
    } else {
        if ((rc = misc_eval_objects_f32(sc, so, (const float*)bx.p, n, (float*)bd.p, (float*)bm.p, (uint8_t*)bh.p, (float*)bc.p, nullptr))) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (d) HIP_TRY(hipMemcpy(d, bd.p, nd * sizeof(R), hipMemcpyDeviceToHost));
    if (dmin) HIP_TRY(hipMemcpy(dmin, bm.p, n * sizeof(R), hipMemcpyDeviceToHost));
    if (hit) HIP_TRY(hipMemcpy(hit, bh.p, n, hipMemcpyDeviceToHost));
    if (rgb) HIP_TRY(hipMemcpy(rgb, bc.p, n * 3 * sizeof(R), hipMemcpyDeviceToHost));
    return RTGR_OK;
}
int api::eval_objects_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* x, uint64_t n, double* d, double* dmin,
                          uint8_t* hit, double* rgb) {
    return eval_objects_host<double>(ctx, scene, opt, x, n, d, dmin, hit, rgb);
}
int api::eval_objects_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* x, uint64_t n, float* d, float* dmin,
                          uint8_t* hit, float* rgb) {
    return eval_objects_host<float>(ctx, scene, opt, x, n, d, dmin, hit, rgb);
}
int api::eval_geodesic_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* s, uint64_t n, int path, double* ds) {
    return eval_geodesic_host<double>(ctx, scene, s, n, path, ds);
}
int api::eval_geodesic_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* s, uint64_t n, int path, float* ds) {
    return eval_geodesic_host<float>(ctx, scene, s, n, path, ds);
}
int api::eval_fastmath_f64(rtgr_context* ctx, const double* x, uint64_t n, double* rcp, double* rsq) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!x) return fail(RTGR_ERR_BAD_ARG, "x is NULL");
    if (n == 0) return RTGR_OK;
    DeviceGuard guard(c->devs[0]->dev);
    DevBuf bx, b1, b2;
    if ((rc = bx.alloc(n * 8))) return rc;
    if (rcp && (rc = b1.alloc(n * 8))) return rc;
    if (rsq && (rc = b2.alloc(n * 8))) return rc;
    HIP_TRY(hipMemcpy(bx.p, x, n * 8, hipMemcpyHostToDevice));
    if ((rc = misc_eval_fastmath_f64((const double*)bx.p, n, (double*)b1.p, (double*)b2.p, nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (rcp) HIP_TRY(hipMemcpy(rcp, b1.p, n * 8, hipMemcpyDeviceToHost));
    if (rsq) HIP_TRY(hipMemcpy(rsq, b2.p, n * 8, hipMemcpyDeviceToHost));
    return RTGR_OK;
}

int api::quantize_device_f64(rtgr_context* ctx, const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, void* stream) {
    if (!d_rgb || !d_img || ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "bad argument");
    RESOLVE_DEVICE(d_rgb);
    DeviceGuard guard(D->dev);
    return misc_quantize(d_rgb, ni, nj, d_img, (hipStream_t)stream);
}

}  // namespace rtgr
