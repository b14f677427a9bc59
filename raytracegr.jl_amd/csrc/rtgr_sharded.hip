// rtgr_sharded.hip — one blocking call over EVERY device of a context, frame assembled in device memory: image rows are
// dealt cyclically, each device traces its rows on a stream of its own, the rows travel to device 0 by peer copies (or through
// pinned host memory where peer access is not available) and are put back in place there.  No kernel here (rtgr_misc.hip: place_rows).
#include "rtgr_internal.hpp"

namespace rtgr {

// ---- all devices of the context ------------------------------------------------------------------------------------------
// per-device scratch of the sharded path lives in the device's Staging: d_out = this rank's rows (all requested arrays),
// d_small = counters; device 0 additionally d_recv = the peers' rows as they arrive.
template <class R>
static int trace_sharded(rtgr_context* c, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni,
                         uint64_t nj, R* d_rgb0, const rtgr_ray_outputs* out0, rtgr_counters* ctr) {
    const uint64_t N = c->devs.size();
    if (!scene || !opt || !cam) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "empty canvas");
    if (c->devs[0]->knobs.tile) return fail(RTGR_ERR_BAD_ARG, "the multi-device path needs the persistent pipeline (option tile = 0)");
    if (out0 && out0->redshift && (!out0->state_end || !(out0->hit || out0->hit32)))
        return fail(RTGR_ERR_BAD_ARG, "rtgr_ray_outputs.redshift needs state_end and hit (or hit32) in the same call");
    struct Arr { size_t elem; int planes; void* full; size_t off; };  // one per requested array
    std::vector<Arr> arrs;
    arrs.push_back({sizeof(R), 3, d_rgb0, 0});
    if (out0) {
        if (out0->state_end) arrs.push_back({8 * sizeof(R), 1, out0->state_end, 0});
        if (out0->lambda_end) arrs.push_back({sizeof(R), 1, out0->lambda_end, 0});
        if (out0->status) arrs.push_back({1, 1, out0->status, 0});
        if (out0->hit) arrs.push_back({1, 1, out0->hit, 0});
        if (out0->n_accept) arrs.push_back({4, 1, out0->n_accept, 0});
        if (out0->n_reject) arrs.push_back({4, 1, out0->n_reject, 0});
        if (out0->redshift) arrs.push_back({sizeof(R), 1, out0->redshift, 0});
        if (out0->hit32) arrs.push_back({4, 1, out0->hit32, 0});
    }
    const uint64_t nrows_max = (nj + N - 1) / N, nmax = ni * nrows_max;
    size_t part_bytes = 0;
    for (auto& a : arrs) { a.off = part_bytes; part_bytes += align256((size_t)nmax * a.elem * a.planes); }
    std::vector<Staging*> S(N, nullptr);
    std::vector<uint64_t> nrows(N, 0);
    std::vector<hipEvent_t> ev(N, nullptr);
    std::vector<char> via_host(N, 0);
    struct EvFree { std::vector<hipEvent_t>& e; ~EvFree() { for (auto x : e) if (x) (void)hipEventDestroy(x); } } evfree{ev};
    int rc;
    // the part / counter / receive buffers are shared by consecutive sharded calls: one such call at a time per context
    // (lock order: device 0's staging mutex first)
    std::vector<std::unique_lock<std::mutex>> locks;
    // Any error return below leaves through this guard FIRST (declared after `locks`, so destroyed before them): once step 1 has
    // started, devices 0..k-1 have kernels and copies in flight on their staging streams that read and write d_out / d_recv /
    // pin_out; returning would drop the Staging locks and destroy the events under them, and the next sharded or host call could
    // reuse or reallocate those buffers beneath running kernels.  So: drain every stream that may have been used.
    struct Drain {
        rtgr_context* c; std::vector<Staging*>& S; bool armed = true;
        ~Drain() {
            if (!armed) return;
            for (size_t k = 0; k < S.size(); k++) {
                if (!S[k]) continue;
                DeviceGuard g(c->devs[k]->dev);
                (void)hipStreamSynchronize(S[k]->s_comp);
                (void)hipStreamSynchronize(S[k]->s_up);
                (void)hipStreamSynchronize(S[k]->s_down);
            }
            (void)hipGetLastError();
        }
    } drain{c, S};
    for (uint64_t k = 0; k < N; k++) {
        DeviceCtx& D = *c->devs[k];
        DeviceGuard g(D.dev);
        { std::lock_guard<std::mutex> lk(D.mu); if ((rc = staging_of(D, &S[k]))) return rc; }
        locks.emplace_back(S[k]->mu);
        nrows[k] = nj > k ? (nj - k + N - 1) / N : 0;
        if ((rc = S[k]->d_out.need(part_bytes))) return rc;
        if ((rc = S[k]->d_small.need(256))) return rc;
        if ((rc = S[k]->pin_small.need(256))) return rc;
        if (k == 0 && N > 1 && (rc = S[0]->d_recv.need(part_bytes * (N - 1)))) return rc;
        HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        // how this device's rows reach device 0: peer copy (default), or D2H + H2D through pinned host memory when peer
        // access could not be enabled at rtgr_create — or when the option peer = 0 forces that path (also between two
        // entries of the SAME physical device: how the fallback is tested on a one-GPU box)
        if (k > 0) {
            const long want = c->devs[0]->knobs.peer;
            if (want == 0) via_host[k] = 1;
            else if (!c->peer_ok[k]) {
                if (want > 0)
                    return fail(RTGR_ERR_HIP, "option peer = 1 but there is no peer access between device " + std::to_string(c->devs[0]->dev) +
                                              " and device " + std::to_string(D.dev) + ": " + c->peer_why[k]);
                via_host[k] = 1;
            }
            if (via_host[k] && (rc = S[k]->pin_out[0].need(part_bytes))) return rc;
        }
    }
    // 1. every device traces its rows on its own stream
    for (uint64_t k = 0; k < N; k++) {
        if (nrows[k] == 0) continue;
        DeviceCtx& D = *c->devs[k];
        DeviceGuard g(D.dev);
        char* pb = (char*)S[k]->d_out.p;
        HIP_TRY(hipMemsetAsync(S[k]->d_small.p, 0, 256, S[k]->s_comp));
        rtgr_ray_outputs po;
        std::memset(&po, 0, sizeof po);
        size_t q = 1;
        if (out0) {
            if (out0->state_end) po.state_end = pb + arrs[q++].off;
            if (out0->lambda_end) po.lambda_end = pb + arrs[q++].off;
            if (out0->status) po.status = (uint8_t*)(pb + arrs[q++].off);
            if (out0->hit) po.hit = (uint8_t*)(pb + arrs[q++].off);
            if (out0->n_accept) po.n_accept = (uint32_t*)(pb + arrs[q++].off);
            if (out0->n_reject) po.n_reject = (uint32_t*)(pb + arrs[q++].off);
            if (out0->redshift) po.redshift = pb + arrs[q++].off;
            if (out0->hit32) po.hit32 = (uint32_t*)(pb + arrs[q++].off);
        }
        rc = trace_device<R>(D, scene, opt, nullptr, cam, ni, nj, k, k + 1, (R*)pb, &po, (rtgr_counters*)S[k]->d_small.p,
                                  S[k]->s_comp, N, nrows[k]);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(S[k]->pin_small.p, S[k]->d_small.p, sizeof(rtgr_counters), hipMemcpyDeviceToHost, S[k]->s_comp));
        // 2. its rows travel to device 0 (ordered behind the trace on the SOURCE device's stream; one xGMI link per peer)
        if (k > 0) {
            char* rb = (char*)S[0]->d_recv.p + (k - 1) * part_bytes;
            const size_t used = part_bytes;  // padded arrays: a single copy per peer
            std::unique_lock<std::mutex> tl(D.mu);   // (the timing list is the device's)
            KernelTimer tm(D, S[k]->s_comp, 4);      // rtgr_timing_read_exchange [0]: this device's rows leaving it
            tl.unlock();
            struct Relock { std::unique_lock<std::mutex>& l; ~Relock() { l.lock(); } } relock{tl};   // ~KernelTimer runs after this: under D.mu again
            if (via_host[k]) HIP_TRY(hipMemcpyAsync(S[k]->pin_out[0].p, pb, used, hipMemcpyDeviceToHost, S[k]->s_comp));
            else if (c->devs[0]->dev == D.dev) HIP_TRY(hipMemcpyAsync(rb, pb, used, hipMemcpyDeviceToDevice, S[k]->s_comp));
            else {
                const hipError_t e = hipMemcpyPeerAsync(rb, c->devs[0]->dev, pb, D.dev, used, S[k]->s_comp);
                if (e != hipSuccess)
                    return fail(RTGR_ERR_HIP, "hipMemcpyPeerAsync device " + std::to_string(D.dev) + " -> device " +
                                              std::to_string(c->devs[0]->dev) + " (" + std::to_string(used) + " bytes): " + hipGetErrorString(e));
            }
        }
        HIP_TRY(hipEventRecord(ev[k], S[k]->s_comp));
    }
    // 2b. rows that travel through the host: wait for the device's D2H (every device has been started by now, so they all
    // run meanwhile), then upload to device 0 on its upload stream; the placement below is ordered behind it by event
    for (uint64_t k = 1; k < N; k++) {
        if (nrows[k] == 0 || !via_host[k]) continue;
        { DeviceGuard g(c->devs[k]->dev);
          const hipError_t e = hipEventSynchronize(ev[k]);
          if (e != hipSuccess) return fail(RTGR_ERR_HIP, "device " + std::to_string(c->devs[k]->dev) + " (rows to the host): " + hipGetErrorString(e)); }
        DeviceGuard g0(c->devs[0]->dev);
        char* rb = (char*)S[0]->d_recv.p + (k - 1) * part_bytes;
        HIP_TRY(hipMemcpyAsync(rb, S[k]->pin_out[0].p, part_bytes, hipMemcpyHostToDevice, S[0]->s_up));
        // "the rows of device k are on device 0": an event of DEVICE 0 (an event is recorded on streams of its own device)
        (void)hipEventDestroy(ev[k]); ev[k] = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev[k], S[0]->s_up));
    }
    // 3. device 0 puts every rank's rows back in place
    {
        DeviceCtx& D0 = *c->devs[0];
        DeviceGuard g(D0.dev);
        hipStream_t s0 = S[0]->s_down;
        for (uint64_t k = 0; k < N; k++) {
            if (nrows[k] == 0) continue;
            HIP_TRY(hipStreamWaitEvent(s0, ev[k], 0));
            std::lock_guard<std::mutex> tl(D0.mu);
            KernelTimer tm(D0, s0, 5);               // rtgr_timing_read_exchange [1]: device 0 putting a rank's rows in place
            const char* src = k == 0 ? (const char*)S[0]->d_out.p : (const char*)S[0]->d_recv.p + (k - 1) * part_bytes;
            for (auto& a : arrs) {
                if (a.planes == 3) {  // rgb: the part's planes are ni*nrows[k] apart
                    if constexpr (sizeof(R) == 8) { if ((rc = misc_place_rows_f64((const double*)(src + a.off), ni, nj, k, N, 3, (double*)a.full, s0))) return rc; }
                    else if ((rc = misc_place_rows_f32((const float*)(src + a.off), ni, nj, k, N, 3, (float*)a.full, s0))) return rc;
                } else if ((rc = misc_place_rows_u8((const uint8_t*)(src + a.off), ni, nj, k, N, a.elem, (uint8_t*)a.full, s0))) return rc;
            }
        }
        HIP_TRY(hipStreamSynchronize(s0));
    }
    rtgr_counters sum;
    std::memset(&sum, 0, sizeof sum);
    for (uint64_t k = 0; k < N; k++) {
        if (nrows[k] == 0) continue;
        DeviceGuard g(c->devs[k]->dev);
        HIP_TRY(hipStreamSynchronize(S[k]->s_comp));
        const uint64_t* p = (const uint64_t*)S[k]->pin_small.p;
        uint64_t* q = (uint64_t*)&sum;
        for (int w = 0; w < 8; w++) q[w] = (w == 7) ? (q[w] > p[w] ? q[w] : p[w]) : q[w] + p[w];   // [7] is a maximum (diagnostics)
    }
    if (ctr) *ctr = sum;
    drain.armed = false;   // every stream used above has been synchronised
    return RTGR_OK;
}

// Host destination: no gather on device 0 is needed — every device downloads its own rows straight into the caller's
// arrays (trace_host_all_devices), which is what rtgr_trace_f64 does on a multi-device context.
template <class R>
static int trace_sharded_host(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni,
                              uint64_t nj, R* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    if (!cam) return fail(RTGR_ERR_BAD_ARG, "camera is NULL");
    return trace_host<R>(ctx, scene, opt, nullptr, cam, ni, nj, 0, nj, rgb, out, ctr);
}
template <class R>
static int trace_sharded_device(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni,
                                uint64_t nj, R* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!d_rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    return trace_sharded<R>(c, scene, opt, cam, ni, nj, d_rgb, out, ctr);
}
int api::trace_sharded_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                                  uint64_t ni, uint64_t nj, double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    return trace_sharded_device<double>(ctx, scene, opt, cam, ni, nj, d_rgb, out, ctr);
}
int api::trace_sharded_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                                  uint64_t ni, uint64_t nj, float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    return trace_sharded_device<float>(ctx, scene, opt, cam, ni, nj, d_rgb, out, ctr);
}
int api::trace_sharded_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                           uint64_t ni, uint64_t nj, double* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    return trace_sharded_host<double>(ctx, scene, opt, cam, ni, nj, rgb, out, ctr);
}
int api::trace_sharded_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                           uint64_t ni, uint64_t nj, float* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    return trace_sharded_host<float>(ctx, scene, opt, cam, ni, nj, rgb, out, ctr);
}

}  // namespace rtgr
