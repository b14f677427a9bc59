// rtgr_host_pipeline.hip — the host-pointer entry points (what a Julia ccall passes): pinned staging and a three-stream
// H2D / compute / D2H pipeline per device, every device of the context driven by a host thread of its own.  No kernel here.
#include "rtgr_internal.hpp"

namespace rtgr {

// ---------------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------
// host <-> pinned copies with a few threads (a single core moves ~8 GB/s; PCIe Gen5 x16 wants ~55)
// ---------------------------------------------------------------------------------------------------------------------
// How many host threads of ONE call are packing / unpacking at the same time (trace_host_all_devices: one per device, each with
// a downloader of its own): the copy threads of a pack are that call's share of the cores, not 8 each — an 8-device context
// would otherwise start up to 8 x 2 x 8 short-lived threads per piece on the same cores and memory channels.
static thread_local unsigned tl_copy_sharers = 1;
template <class F>
void parallel_rows(uint64_t n, size_t bytes_per_item, F&& body) {  // body(first, count)
    const size_t total = (size_t)n * bytes_per_item;
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = total < (4u << 20) ? 1u : (hw >= 16 ? 8u : (hw >= 4 ? hw / 2 : 1u));
    if (tl_copy_sharers > 1) {   // this call's share of the cores: (cores / 2 sharers), at least one thread
        const unsigned share = hw / (2 * tl_copy_sharers);
        nt = nt < (share ? share : 1u) ? nt : (share ? share : 1u);
    }
    if (nt <= 1) { body((uint64_t)0, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t a = (uint64_t)t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        th.emplace_back([&, a, b] { body(a, b - a); });
    }
    for (auto& t : th) t.join();
}

// One output array of a host-pointer call: `planes` planes (rgb: 3, else 1) of `elem` bytes per ray.
struct OutArray { void* host; size_t elem; int planes; size_t dev_off; };

// The host-pointer hot path (rtgr_trace_f64 / _f32 / _pixels_f64 / _one_f64).  Rays [0, n) are rows [j0, j1) of the
// canvas; the job is cut into compute chunks of whole rows (~2^22 rays: big enough that the persistent kernels lose
// nothing, SURVEY §6 / DESIGN §4.2) and each chunk's input into transfer pieces (~2^20 rays).  Three streams:
//     s_up:   H2D of the pieces from pinned staging (the host packs them there with a few threads)
//     s_comp: the trace pipeline of chunk c once its last piece has landed
//     s_down: D2H of chunk c's outputs into pinned staging; a helper thread unpacks them into the caller's arrays
// so that upload, integration and download of successive chunks overlap, and PCIe carries 64 B/ray in (ray states; nothing
// when the camera generates them on the device) and 24 B/ray out (+ what `out` asks for) — never the 88-byte pixels.
// `px_in` != NULL: the input is the reference's Pixel{T} array (11 scalars per pixel, pos + normal are packed
// out of it on the way up) and `px_out` receives Pixel(p.pos, p.normal, rgb) (:532).
// `rows` (multi-device contexts): this device's share of the slab — local row k is slab row rows.first + k * rows.stride
// (cyclic rows).  The device buffers hold the LOCAL rays contiguously; the row map is applied where rays are
// packed out of / unpacked into the caller's arrays, so every device reads its rows from, and writes them straight back
// into, the caller's host memory over its own PCIe link — no hop through device 0.
struct RowShare { uint64_t first = 0, stride = 1; };

template <class R>
int trace_host_pipelined(DeviceCtx& D, const rtgr_scene* scene, const rtgr_solver* opt, const R* state0, const R* px_in,
                         R* px_out, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* rgb,
                         const rtgr_ray_outputs* out, rtgr_counters* ctr, RowShare share = RowShare(), int slot = 0) {
    DeviceGuard guard(D.dev);
    if (!guard.ok) return fail(RTGR_ERR_HIP, "hipSetDevice failed");
    Staging* S = nullptr;
    int rc;
    { std::lock_guard<std::mutex> lk(D.mu); if ((rc = staging_of(D, &S, slot))) return rc; }
    std::lock_guard<std::mutex> call_lock(S->mu);
    // a previous call that failed half-way may have left copies in flight on the staging streams: they are idle otherwise
    HIP_TRY(hipStreamSynchronize(S->s_up));
    HIP_TRY(hipStreamSynchronize(S->s_comp));
    HIP_TRY(hipStreamSynchronize(S->s_down));
    const uint64_t nrows_slab = j1 - j0, n_slab = ni * nrows_slab;   // the caller's arrays
    const uint64_t nrows = nrows_slab > share.first ? (nrows_slab - share.first + share.stride - 1) / share.stride : 0;
    const uint64_t n = ni * nrows;                                     // this device's rays
    if (ctr) std::memset(ctr, 0, sizeof *ctr);
    if (n == 0) return RTGR_OK;
    const bool have_in = state0 != nullptr || px_in != nullptr;
    const bool strided = share.stride != 1;
    // local rays [a, a + cnt) as runs of consecutive rays of the caller's arrays: body(local_first, global_first, length)
    auto for_runs = [ni, share, strided](uint64_t a, uint64_t cnt, auto&& body) {
        if (!strided) { body(a, share.first * ni + a, cnt); return; }
        while (cnt > 0) {
            const uint64_t k = a / ni, i = a % ni;
            const uint64_t len = (ni - i) < cnt ? (ni - i) : cnt;
            body(a, (share.first + k * share.stride) * ni + i, len);
            a += len; cnt -= len;
        }
    };

    // ---- output arrays ---------------------------------------------------------------------------------------------
    std::vector<OutArray> outs;
    size_t dev_bytes = 0;
    auto add = [&](void* host, size_t elem, int planes) {
        outs.push_back({host, elem, planes, dev_bytes});
        dev_bytes += align256((size_t)n * elem * planes);
    };
    add(px_in ? nullptr : (void*)rgb, sizeof(R), 3);  // [0] = rgb, always (pixels: unpacked into px_out)
    rtgr_ray_outputs dout;
    std::memset(&dout, 0, sizeof dout);
    if (out) {
        if (out->redshift && (!out->state_end || !(out->hit || out->hit32)))
            return fail(RTGR_ERR_BAD_ARG, "rtgr_ray_outputs.redshift needs state_end and hit (or hit32) in the same call");
        if (out->state_end) add(out->state_end, 8 * sizeof(R), 1);
        if (out->lambda_end) add(out->lambda_end, sizeof(R), 1);
        if (out->status) add(out->status, 1, 1);
        if (out->hit) add(out->hit, 1, 1);
        if (out->n_accept) add(out->n_accept, 4, 1);
        if (out->n_reject) add(out->n_reject, 4, 1);
        if (out->redshift) add(out->redshift, sizeof(R), 1);
        if (out->hit32) add(out->hit32, 4, 1);
    }
    size_t out_bytes_per_ray = 0;
    for (auto& o : outs) out_bytes_per_ray += o.elem * o.planes;

    // ---- chunking ----------------------------------------------------------------------------------------------------
    // Compute chunks of whole rows, sized P, 2P, 4P, 4P, …, 4P, P (P = one transfer piece, ~2^20 rays): the FIRST chunk
    // is small so that integration starts after one piece has been packed and uploaded instead of four (the pipeline's
    // fill), the LAST so that only a small unpack follows the last kernel (its drain); in between the chunks are big
    // enough (4 M rays) that the persistent kernels lose nothing to their tails.  Measured at 4096² through
    // rtgr_trace_pixels_f64: 119 ms with equal 4 M-ray chunks, 106-112 ms with the ramp, 101-103 ms with the D2H ordered behind the
    // next chunk's set-up (device-resident: 86-88 ms;
    // measured split of a 106 ms call: chunked compute 95, fill 5, drain 3, the rest host noise).  Measured and rejected:
    // alternating the chunks between TWO compute streams so that their tails overlap — the persistent kernels of two
    // pipelines in flight slow each other down more than the tails cost (pixels 109 -> 123 ms, host 91 -> 98 ms; re-measured
    // with event-only dependencies: 114 -> 118, 95 -> 98).  Two whole FRAMES in flight on two caller streams do pay — equal,
    // independent jobs (tools/two_frames_in_flight.py) — but that is the caller's loop, not this call's.
    const uint64_t piece_target = D.knobs.host_chunk > 0 ? (uint64_t)D.knobs.host_chunk : (1ull << 20);
    struct Chunk { uint64_t row0, rows; };
    std::vector<Chunk> chunks;
    if (D.knobs.tile) {
        chunks.push_back({0, nrows});   // (the tile kernel writes whole slabs)
    } else {
        auto rows_for = [&](uint64_t rays) { const uint64_t r = rays / ni; return r ? r : (uint64_t)1; };
        // (middle chunks: 4P, or an eighth of a big job — 8192² with the disk loses 10 % of its FAR pass to the tails of
        //  sixteen 4 M-ray launches, nothing to those of eight 8 M-ray ones)
        const uint64_t mid = 4 * piece_target > n / 8 ? 4 * piece_target : n / 8;
        const uint64_t rP = rows_for(piece_target), r2 = rows_for(2 * piece_target), r4 = rows_for(mid);
        std::vector<uint64_t> head, tail;
        uint64_t left = nrows;
        const uint64_t ramp[2] = {rP, r2};
        for (int k = 0; k < 2 && left > 0; k++) {            // P, 2P from the front (only when there is input to wait for) ...
            if (have_in) {
                const uint64_t h = ramp[k] < left ? ramp[k] : left;
                head.push_back(h); left -= h;
            }
            if (left == 0) break;
            if (have_in && k > 0) continue;                       // ... and from the back: P (and 2P when nothing is uploaded)
            const uint64_t t = ramp[k] < left ? ramp[k] : left;
            tail.push_back(t); left -= t;
        }
        if (left > 0) {                                       // the middle, in equal chunks of at most r4 rows — ONE chunk when
            // there is no input to wait for (camera on the device): every extra launch costs its tails (~1.2 ms at 4 M rays),
            // and the only reason to cut at all is to hide the last download behind the small chunks at the end
            const uint64_t parts = have_in ? (left + r4 - 1) / r4 : 1;
            for (uint64_t k = 0; k < parts; k++) {
                const uint64_t m4 = (left + (parts - k) - 1) / (parts - k);
                head.push_back(m4); left -= m4;
            }
        }
        uint64_t row0 = 0;
        for (uint64_t r : head) { chunks.push_back({row0, r}); row0 += r; }
        for (size_t k = tail.size(); k-- > 0;) { chunks.push_back({row0, tail[k]}); row0 += tail[k]; }
    }
    const uint64_t nchunks = chunks.size();
    uint64_t chunk_rays_max = 0;
    for (auto& ch : chunks) chunk_rays_max = ch.rows * ni > chunk_rays_max ? ch.rows * ni : chunk_rays_max;
    const uint64_t piece = chunk_rays_max < piece_target ? chunk_rays_max : piece_target;

    // ---- buffers (grow-only; device streams of the staging are idle here: every call ends synchronised) ----------------
    if (have_in) {
        if ((rc = S->d_in.need((size_t)n * 8 * sizeof(R)))) return rc;
        for (auto& b : S->pin_in) if ((rc = b.need((size_t)piece * 8 * sizeof(R)))) return rc;
    }
    if ((rc = S->d_out.need(dev_bytes))) return rc;
    for (auto& b : S->pin_out) if ((rc = b.need((size_t)chunk_rays_max * out_bytes_per_ray + 256 * outs.size() * 3))) return rc;
    if ((rc = S->d_small.need(256))) return rc;
    if ((rc = S->pin_small.need(256))) return rc;
    char* dsmall = (char*)S->d_small.p;
    rtgr_counters* d_ctr = (rtgr_counters*)dsmall;
    uint32_t* d_nan = (uint32_t*)(dsmall + 128);
    HIP_TRY(hipMemsetAsync(dsmall, 0, 256, S->s_comp));
    char* dob = (char*)S->d_out.p;
    R* d_rgb = (R*)(dob + outs[0].dev_off);
    {
        size_t k = 1;
        if (out) {
            if (out->state_end) dout.state_end = dob + outs[k++].dev_off;
            if (out->lambda_end) dout.lambda_end = dob + outs[k++].dev_off;
            if (out->status) dout.status = (uint8_t*)(dob + outs[k++].dev_off);
            if (out->hit) dout.hit = (uint8_t*)(dob + outs[k++].dev_off);
            if (out->n_accept) dout.n_accept = (uint32_t*)(dob + outs[k++].dev_off);
            if (out->n_reject) dout.n_reject = (uint32_t*)(dob + outs[k++].dev_off);
            if (out->redshift) dout.redshift = dob + outs[k++].dev_off;
            if (out->hit32) dout.hit32 = (uint32_t*)(dob + outs[k++].dev_off);
        }
    }
    R* d_in = (R*)S->d_in.p;

    std::vector<hipEvent_t> ev_comp(nchunks), ev_down(nchunks), ev_setup(nchunks);
    for (auto& e : ev_comp) { e = nullptr; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    for (auto& e : ev_down) { e = nullptr; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    for (auto& e : ev_setup) { e = nullptr; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    struct EvFree { std::vector<hipEvent_t>&a, &b, &c; ~EvFree() { for (auto* v : {&a, &b, &c}) for (auto e : *v) if (e) (void)hipEventDestroy(e); } } evfree{ev_comp, ev_down, ev_setup};
    hipEvent_t ev_up_last = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev_up_last, hipEventDisableTiming));
    struct OneEv { hipEvent_t e; ~OneEv() { if (e) (void)hipEventDestroy(e); } } onefree{ev_up_last};

    // ---- download side: a helper thread waits for each chunk's D2H and unpacks it into the caller's arrays --------------
    std::atomic<int> down_rc{RTGR_OK};
    std::atomic<uint64_t> chunks_enqueued{0};
    std::atomic<bool> abort_flag{false};
    std::vector<std::atomic<int>> slot_busy(Staging::OUT_SLOTS);
    for (auto& b : slot_busy) b.store(0);
    auto chunk_range = [&](uint64_t c, uint64_t& r0, uint64_t& m) { r0 = chunks[c].row0 * ni; m = chunks[c].rows * ni; };
    const int dev_ordinal = D.dev;
    const unsigned copy_sharers = tl_copy_sharers;
    std::thread downloader([&] {
        (void)hipSetDevice(dev_ordinal);
        tl_copy_sharers = copy_sharers;
        for (uint64_t c = 0; c < nchunks; c++) {
            while (chunks_enqueued.load(std::memory_order_acquire) <= c) {
                if (abort_flag.load()) return;
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (hipEventSynchronize(ev_down[c]) != hipSuccess) { down_rc.store(RTGR_ERR_HIP); return; }
            uint64_t r0, m;
            chunk_range(c, r0, m);
            const char* src = (const char*)S->pin_out[c % Staging::OUT_SLOTS].p;
            size_t off = 0;
            for (size_t k = 0; k < outs.size(); k++) {
                const OutArray& o = outs[k];
                if (k == 0 && px_in) {  // Pixel(p.pos, p.normal, col)  (:532)
                    const R* pr = (const R*)(src + off);
                    parallel_rows(m, 11 * sizeof(R), [&](uint64_t a, uint64_t cnt) {
                        for_runs(r0 + a, cnt, [&](uint64_t l0, uint64_t g0, uint64_t len) {
                            for (uint64_t e = 0; e < len; e++) {
                                const uint64_t w = l0 - r0 + e;
                                R* po = px_out + (g0 + e) * 11;
                                const R* pi = px_in + (g0 + e) * 11;
                                if (po != pi) for (int q = 0; q < 8; q++) po[q] = pi[q];
                                po[8] = pr[w]; po[9] = pr[m + w]; po[10] = pr[2 * m + w];
                            }
                        });
                    });
                } else {
                    for (int pl = 0; pl < o.planes; pl++) {
                        char* dst = (char*)o.host + (size_t)pl * n_slab * o.elem;
                        const char* s2 = src + off + (size_t)pl * m * o.elem;
                        parallel_rows(m, o.elem, [&](uint64_t a, uint64_t cnt) {
                            for_runs(r0 + a, cnt, [&](uint64_t l0, uint64_t g0, uint64_t len) {
                                std::memcpy(dst + g0 * o.elem, s2 + (l0 - r0) * o.elem, len * o.elem);
                            });
                        });
                    }
                }
                off += align256((size_t)m * o.elem * o.planes);
            }
            slot_busy[c % Staging::OUT_SLOTS].store(0, std::memory_order_release);
        }
    });
    struct Joiner { std::thread& t; std::atomic<bool>& ab; ~Joiner() { ab.store(true); if (t.joinable()) t.join(); } } joiner{downloader, abort_flag};

    // D2H of chunk c's outputs into its pinned slot (the slot must have been unpacked: two chunks ago)
    auto enqueue_download = [&](uint64_t c, hipEvent_t also_after) -> int {
        uint64_t r0, m;
        chunk_range(c, r0, m);
        const int oslot = (int)(c % Staging::OUT_SLOTS);
        while (slot_busy[oslot].load(std::memory_order_acquire)) {
            if (down_rc.load() != RTGR_OK) return fail(RTGR_ERR_HIP, "download thread failed");
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        slot_busy[oslot].store(1);
        HIP_TRY(hipStreamWaitEvent(S->s_down, ev_comp[c], 0));
        if (also_after) HIP_TRY(hipStreamWaitEvent(S->s_down, also_after, 0));
        char* dst = (char*)S->pin_out[oslot].p;
        size_t off = 0;
        for (auto& o : outs) {
            for (int pl = 0; pl < o.planes; pl++)
                HIP_TRY(hipMemcpyAsync(dst + off + (size_t)pl * m * o.elem, dob + o.dev_off + ((size_t)pl * n + r0) * o.elem,
                                       (size_t)m * o.elem, hipMemcpyDeviceToHost, S->s_down));
            off += align256((size_t)m * o.elem * o.planes);
        }
        HIP_TRY(hipEventRecord(ev_down[c], S->s_down));
        chunks_enqueued.store(c + 1, std::memory_order_release);
        return RTGR_OK;
    };

    // ---- upload + compute, chunk by chunk ---------------------------------------------------------------------------------
    uint64_t piece_no = 0;
    for (uint64_t c = 0; c < nchunks; c++) {
        uint64_t r0, m;
        chunk_range(c, r0, m);
        if (have_in) {
            for (uint64_t p0 = 0; p0 < m; p0 += piece, piece_no++) {
                const uint64_t pm = (m - p0) < piece ? (m - p0) : piece;
                const int slot = (int)(piece_no % Staging::IN_SLOTS);
                if (piece_no >= (uint64_t)Staging::IN_SLOTS) HIP_TRY(hipEventSynchronize(S->ev_in[slot]));  // slot's last H2D done
                R* pin = (R*)S->pin_in[slot].p;
                const uint64_t l00 = r0 + p0;   // first local ray of the piece
                if (px_in) {
                    parallel_rows(pm, 8 * sizeof(R), [&](uint64_t a, uint64_t cnt) {
                        for_runs(l00 + a, cnt, [&](uint64_t l0, uint64_t g0, uint64_t len) {
                            for (uint64_t e = 0; e < len; e++) {
                                const R* pi = px_in + (g0 + e) * 11;
                                R* d = pin + (l0 - l00 + e) * 8;
                                for (int q = 0; q < 8; q++) d[q] = pi[q];
                            }
                        });
                    });
                } else {
                    parallel_rows(pm, 8 * sizeof(R), [&](uint64_t a, uint64_t cnt) {
                        for_runs(l00 + a, cnt, [&](uint64_t l0, uint64_t g0, uint64_t len) {
                            std::memcpy(pin + (l0 - l00) * 8, state0 + g0 * 8, len * 8 * sizeof(R));
                        });
                    });
                }
                HIP_TRY(hipMemcpyAsync(d_in + l00 * 8, pin, (size_t)pm * 8 * sizeof(R), hipMemcpyHostToDevice, S->s_up));
                HIP_TRY(hipEventRecord(S->ev_in[slot], S->s_up));
            }
            HIP_TRY(hipEventRecord(ev_up_last, S->s_up));
            HIP_TRY(hipStreamWaitEvent(S->s_comp, ev_up_last, 0));
        }
        Window win;
        win.plane_stride = n; win.out_offset = r0; win.nan_flag = have_in ? d_nan : nullptr;
        win.after_setup = ev_setup[c];
        const uint64_t row0 = chunks[c].row0, rows = chunks[c].rows;
        if (D.knobs.tile) { win.plane_stride = 0; win.out_offset = 0; }
        if (!strided)
            rc = trace_device<R>(D, scene, opt, have_in ? d_in + r0 * 8 : nullptr, cam, ni, nj, j0 + share.first + row0,
                                 j0 + share.first + row0 + rows, d_rgb, &dout, d_ctr, S->s_comp, 1, 0, &win);
        else {  // local rows row0 … of a cyclic share: image rows j0 + first + (row0 + k) * stride
            const uint64_t jf = j0 + share.first + row0 * share.stride;
            rc = trace_device<R>(D, scene, opt, have_in ? d_in + r0 * 8 : nullptr, cam, ni, nj, jf, jf + 1, d_rgb, &dout, d_ctr,
                                 S->s_comp, share.stride, rows, &win);
        }
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ev_comp[c], S->s_comp));
        // The D2H of chunk c-1 goes out only now, behind the SET-UP kernels of chunk c: the runtime copies device -> host
        // with blit kernels, and the memory-bound set-up kernels crawl next to them (rocprofv3 timeline at 4096²: prepare of a
        // 4 M-ray chunk 1.9 ms beside the copies, 0.3 ms alone); beside the VALU-bound integrate pass they cost nothing.
        if (c > 0 && (rc = enqueue_download(c - 1, ev_setup[c]))) return rc;
    }
    if ((rc = enqueue_download(nchunks - 1, nullptr))) return rc;
    // counters + NaN flag ride the compute stream
    HIP_TRY(hipMemcpyAsync(S->pin_small.p, dsmall, 256, hipMemcpyDeviceToHost, S->s_comp));
    HIP_TRY(hipStreamSynchronize(S->s_comp));
    downloader.join();
    HIP_TRY(hipStreamSynchronize(S->s_down));
    HIP_TRY(hipStreamSynchronize(S->s_up));
    if (down_rc.load() != RTGR_OK) return fail(RTGR_ERR_HIP, "download thread failed");
    if (*(const uint32_t*)((const char*)S->pin_small.p + 128) != 0u)
        return fail(RTGR_ERR_NAN_INPUT, "NaN in an input ray (AssertionError in the reference, :279)");
    if (ctr) std::memcpy(ctr, S->pin_small.p, sizeof(rtgr_counters));
    return RTGR_OK;
}

// The host-pointer hot path over EVERY device of the context — what `trace_rays(metric, objs, canvas)` binds
// (src/RayTraceGR.jl:483-536; call sites :560, :596).  Rows of the slab are dealt cyclically: device k of N takes slab rows
// k, k+N, …, runs the three-stream pipeline above on them from a host thread of its own, reads ITS rows from the caller's
// array and writes them straight back (H2D and D2H over the device's own PCIe link; nothing is routed through device 0);
// counters are summed.  One device (or a one-row slab): the plain single-device call on the calling thread.
template <class R>
int trace_host_all_devices(rtgr_context* c, const rtgr_scene* scene, const rtgr_solver* opt, const R* state0, const R* px_in,
                           R* px_out, const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* rgb,
                           const rtgr_ray_outputs* out, rtgr_counters* ctr, int slot = 0) {
    const uint64_t nrows = j1 - j0;
    const uint64_t N = c->devs.size() < nrows ? c->devs.size() : nrows;
    if (N <= 1) return trace_host_pipelined<R>(*c->devs[0], scene, opt, state0, px_in, px_out, cam, ni, nj, j0, j1, rgb, out, ctr, RowShare(), slot);
    if (c->devs[0]->knobs.tile) return fail(RTGR_ERR_BAD_ARG, "the multi-device path needs the persistent pipeline (option tile = 0)");
    std::vector<int> rcs(N, RTGR_OK);
    std::vector<std::string> errs(N);
    std::vector<rtgr_counters> ctrs(N);
    auto work = [&](uint64_t k) {
        RowShare sh;
        sh.first = k; sh.stride = N;
        struct Sharers { unsigned prev; explicit Sharers(unsigned n) : prev(tl_copy_sharers) { tl_copy_sharers = n; } ~Sharers() { tl_copy_sharers = prev; } } sharers((unsigned)N);
        rcs[k] = trace_host_pipelined<R>(*c->devs[k], scene, opt, state0, px_in, px_out, cam, ni, nj, j0, j1, rgb, out, &ctrs[k], sh, slot);
        if (rcs[k]) errs[k] = last_error_string();   // the message is per thread: carry it to the caller's
    };
    {
        std::vector<std::thread> th;
        for (uint64_t k = 1; k < N; k++) th.emplace_back(work, k);
        work(0);
        for (auto& t : th) t.join();
    }
    for (uint64_t k = 0; k < N; k++)
        if (rcs[k]) return fail(rcs[k], "device " + std::to_string(c->devs[k]->dev) + " (entry " + std::to_string(k) + " of the context): " + errs[k]);
    if (ctr) {
        std::memset(ctr, 0, sizeof *ctr);
        for (uint64_t k = 0; k < N; k++) {
            const uint64_t* p = (const uint64_t*)&ctrs[k];
            uint64_t* q = (uint64_t*)ctr;
            for (int w = 0; w < 8; w++) q[w] = (w == 7) ? (q[w] > p[w] ? q[w] : p[w]) : q[w] + p[w];   // [7] is a maximum (diagnostics)
        }
    }
    return RTGR_OK;
}

template <class R>
int trace_host(rtgr_context* ctx_in, const rtgr_scene* scene, const rtgr_solver* opt, const R* state0, const rtgr_camera* cam,
               uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx_in, &c);
    if (rc) return rc;
    if (!rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    if (ni == 0 || nj == 0 || j1 <= j0 || j1 > nj) return fail(RTGR_ERR_BAD_ARG, "bad canvas range: need 0 <= j0 < j1 <= nj, ni > 0");
    if (!state0 && !cam) return fail(RTGR_ERR_BAD_ARG, "need state0 or a camera");
    if (!scene) return fail(RTGR_ERR_BAD_ARG, "scene is NULL");
    return trace_host_all_devices<R>(c, scene, opt, state0, nullptr, nullptr, cam, ni, nj, j0, j1, rgb, out, ctr);
}

template int trace_host<double>(rtgr_context*, const rtgr_scene*, const rtgr_solver*, const double*, const rtgr_camera*, uint64_t, uint64_t, uint64_t,
                                uint64_t, double*, const rtgr_ray_outputs*, rtgr_counters*);
template int trace_host<float>(rtgr_context*, const rtgr_scene*, const rtgr_solver*, const float*, const rtgr_camera*, uint64_t, uint64_t, uint64_t,
                               uint64_t, float*, const rtgr_ray_outputs*, rtgr_counters*);

int api::trace_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* state0,
                   const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* rgb,
                   const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    return trace_host<double>(ctx, scene, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr);
}
int api::trace_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* state0,
                   const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* rgb,
                   const rtgr_ray_outputs* out, rtgr_counters* ctr) {
    return trace_host<float>(ctx, scene, opt, state0, cam, ni, nj, j0, j1, rgb, out, ctr);
}

template <class R>
static int trace_pixels(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const R* pixels_in, uint64_t ni,
                        uint64_t nj, R* pixels_out, rtgr_counters* ctr) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!pixels_in || !pixels_out) return fail(RTGR_ERR_BAD_ARG, "pixels is NULL");
    if (ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "empty canvas");
    if (!scene) return fail(RTGR_ERR_BAD_ARG, "scene is NULL");
    return trace_host_all_devices<R>(c, scene, opt, nullptr, pixels_in, pixels_out, nullptr, ni, nj, 0, nj, nullptr, nullptr, ctr);
}
template <class R>
static int trace_one(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const R pos[4], const R normal[4],
                     R rgb[3], R state_end[8], uint8_t* status) {
    if (!pos || !normal || !rgb) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    R s0[8];
    for (int q = 0; q < 4; q++) { s0[q] = pos[q]; s0[4 + q] = normal[q]; }
    rtgr_ray_outputs out;
    std::memset(&out, 0, sizeof out);
    out.state_end = state_end;
    out.status = status;
    return trace_host<R>(ctx, scene, opt, s0, nullptr, 1, 1, 0, 1, rgb, &out, nullptr);
}
int api::trace_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* pixels_in,
                          uint64_t ni, uint64_t nj, double* pixels_out, rtgr_counters* ctr) {
    return trace_pixels<double>(ctx, scene, opt, pixels_in, ni, nj, pixels_out, ctr);
}
int api::trace_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* pixels_in,
                          uint64_t ni, uint64_t nj, float* pixels_out, rtgr_counters* ctr) {
    return trace_pixels<float>(ctx, scene, opt, pixels_in, ni, nj, pixels_out, ctr);
}
int api::trace_one_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double pos[4],
                       const double normal[4], double rgb[3], double state_end[8], uint8_t* status) {
    return trace_one<double>(ctx, scene, opt, pos, normal, rgb, state_end, status);
}
int api::trace_one_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float pos[4],
                       const float normal[4], float rgb[3], float state_end[8], uint8_t* status) {
    return trace_one<float>(ctx, scene, opt, pos, normal, rgb, state_end, status);
}

// ---------------------------------------------------------------------------------------------------------------------
// several frames in one call, two in flight (an extension: the reference renders one frame per call, :560, :596)
// ---------------------------------------------------------------------------------------------------------------------
// A render loop delivers frame after frame, and every frame's pipeline ends thin: the last rays of its FAR pass, the long stayers of
// its NEAR pass, the download of its last chunk.  With TWO frames in flight — each on a pipeline of its own: staging buffers, three
// streams, the per-stream workspace — the thin end of one overlaps the start of the next (measured through the device entry on two
// caller streams: 1024² a = 0 6.67 -> 6.2 ms per frame, a = 0.8 9.53 -> 8.4; an N = 8 share of 4096² 12.1 -> 11.5).  A
// caller that owns HIP streams could always do that; a Julia or C caller of the blocking entry points could not — this call does
// it for them: frames 0, 2, 4, … run on the calling thread through the context's first pipeline, frames 1, 3, 5, … on a helper
// thread through its second, every frame over ALL devices of the context exactly as rtgr_trace_f64 / rtgr_trace_pixels_f64 deal
// it.  Frame k's results are those of the single call, bit for bit (tests).
template <class R>
static int trace_frames(rtgr_context* ctx_in, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams,
                        const R* const* state0s, const R* const* px_in, R* const* px_out, uint64_t ni, uint64_t nj, R* const* rgb,
                        const rtgr_ray_outputs* outs, rtgr_counters* ctrs) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx_in, &c);
    if (rc) return rc;
    if (!scene || !opt) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (nframes == 0) return fail(RTGR_ERR_BAD_ARG, "no frames");
    if (ni == 0 || nj == 0) return fail(RTGR_ERR_BAD_ARG, "empty canvas");
    if (px_in) {
        if (!px_out) return fail(RTGR_ERR_BAD_ARG, "pixels_out is NULL");
    } else {
        if (!rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
        if (!cams && !state0s) return fail(RTGR_ERR_BAD_ARG, "need cameras or ray states");
    }
    for (uint32_t k = 0; k < nframes; k++) {
        if (px_in ? (!px_in[k] || !px_out[k]) : !rgb[k]) return fail(RTGR_ERR_BAD_ARG, "frame " + std::to_string(k) + ": NULL array");
        if (!px_in && !cams && !state0s[k]) return fail(RTGR_ERR_BAD_ARG, "frame " + std::to_string(k) + ": NULL ray states");
    }
    auto one = [&](uint32_t k, int slot) {
        return trace_host_all_devices<R>(c, scene, opt, (state0s && !px_in) ? state0s[k] : nullptr, px_in ? px_in[k] : nullptr,
                                         px_in ? px_out[k] : nullptr, (cams && !px_in && !(state0s && state0s[k])) ? &cams[k] : nullptr, ni, nj, 0, nj,
                                         px_in ? nullptr : rgb[k], (outs && !px_in) ? &outs[k] : nullptr, ctrs ? &ctrs[k] : nullptr, slot);
    };
    int rc2 = RTGR_OK;
    uint32_t bad2 = 0;
    std::string err2;
    std::atomic<bool> stop{false};
    std::thread second;
    if (nframes > 1)
        second = std::thread([&] {
            for (uint32_t k = 1; k < nframes && !stop.load(); k += 2)
                if ((rc2 = one(k, 1))) { bad2 = k; err2 = last_error_string(); stop.store(true); break; }
        });
    uint32_t bad = 0;
    for (uint32_t k = 0; k < nframes && !stop.load(); k += 2)
        if ((rc = one(k, 0))) { bad = k; stop.store(true); break; }
    if (second.joinable()) second.join();
    if (rc) return fail(rc, "frame " + std::to_string(bad) + ": " + last_error_string());
    if (rc2) return fail(rc2, "frame " + std::to_string(bad2) + ": " + err2);
    return RTGR_OK;
}
int api::trace_frames_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams,
                          const double* const* state0s, uint64_t ni, uint64_t nj, double* const* rgb, const rtgr_ray_outputs* outs, rtgr_counters* ctrs) {
    return trace_frames<double>(ctx, scene, opt, nframes, cams, state0s, nullptr, nullptr, ni, nj, rgb, outs, ctrs);
}
int api::trace_frames_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams,
                          const float* const* state0s, uint64_t ni, uint64_t nj, float* const* rgb, const rtgr_ray_outputs* outs, rtgr_counters* ctrs) {
    return trace_frames<float>(ctx, scene, opt, nframes, cams, state0s, nullptr, nullptr, ni, nj, rgb, outs, ctrs);
}
int api::trace_frames_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const double* const* pixels_in,
                                 uint64_t ni, uint64_t nj, double* const* pixels_out, rtgr_counters* ctrs) {
    return trace_frames<double>(ctx, scene, opt, nframes, nullptr, nullptr, pixels_in, pixels_out, ni, nj, nullptr, nullptr, ctrs);
}
int api::trace_frames_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const float* const* pixels_in,
                                 uint64_t ni, uint64_t nj, float* const* pixels_out, rtgr_counters* ctrs) {
    return trace_frames<float>(ctx, scene, opt, nframes, nullptr, nullptr, pixels_in, pixels_out, ni, nj, nullptr, nullptr, ctrs);
}

}  // namespace rtgr
