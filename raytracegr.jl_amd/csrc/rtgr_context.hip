// rtgr_context.hip — contexts and the device entry points of include/rtgr.h (gfx950 only; no CPU fallback, no compatibility paths).
//
// This file holds NO kernel: the calling thread's error string, launch options, contexts and their per-device / per-stream state
// (rtgr_host.hpp), argument checking and conversion, and the enqueue of one trace on one device — dispatch to the translation unit
// that owns the metric variant's kernels (tu_*.hip).  The other host units build on it (rtgr_internal.hpp lists them).
#include "rtgr_internal.hpp"

namespace rtgr {

// ---------------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// ---------------------------------------------------------------------------------------------------------------------
// knobs
// ---------------------------------------------------------------------------------------------------------------------
static const char* const KNOB_NAMES[] = {"waves_per_cu", "waves_per_cu_near", "chunk", "split", "order", "fair", "near_early",
                                         "far4", "rounds", "qchunk", "qchunk_near", "tile", "host_chunk",
                                         "dbg_pass_far", "peer", "pack", "packfar", "unit_probe", "unit_audit", "max_waves", "scene_check", "handback_after", "groups", nullptr};
const char* const* knob_names() { return KNOB_NAMES; }
long* knob_slot(Knobs& k, const char* name) {
    if (!name) return nullptr;
    long* slots[] = {&k.waves_per_cu, &k.waves_per_cu_near, &k.chunk, &k.split, &k.order, &k.fair, &k.near_early,
                     &k.far4, &k.rounds, &k.qchunk, &k.qchunk_near, &k.tile, &k.host_chunk, &k.dbg_pass_far, &k.peer, &k.pack, &k.packfar,
                     &k.unit_probe, &k.unit_audit, &k.max_waves, &k.scene_check, &k.handback_after, &k.groups};
    for (int i = 0; KNOB_NAMES[i]; i++)
        if (std::strcmp(KNOB_NAMES[i], name) == 0) return slots[i];
    return nullptr;
}
static Knobs knobs_from_env() {  // once per context
    Knobs k;
    for (int i = 0; KNOB_NAMES[i]; i++) {
        std::string env = "RTGR_";
        for (const char* c = KNOB_NAMES[i]; *c; c++) env += (char)std::toupper((unsigned char)*c);
        const char* v = std::getenv(env.c_str());
        if (v && *v) *knob_slot(k, KNOB_NAMES[i]) = std::atol(v);
    }
    const char* kn = std::getenv("RTGR_KERNEL");  // historical spelling of tile = 1
    if (kn && std::strcmp(kn, "tile") == 0) k.tile = 1;
    return k;
}

// ---------------------------------------------------------------------------------------------------------------------
// workspaces
// ---------------------------------------------------------------------------------------------------------------------
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

template <class R>
size_t workspace_bytes(uint64_t rays, bool with_state) {
    const int recw = with_state ? REC_TAIL_STATE : REC_TAIL;   // the event records' tails; their heads overlay the hand-over records
    static_assert(HAND_W <= REC_W, "an event record's head overlays the ray's hand-over record");
    return align256(rays * recw * sizeof(R)) + align256(rays * HAND_W * sizeof(R)) + align256(rays * 3 * sizeof(uint32_t)) +
           2 * align256(rays * sizeof(uint32_t)) + align256(rays) + 4096;
}
template size_t workspace_bytes<double>(uint64_t, bool);
template size_t workspace_bytes<float>(uint64_t, bool);

// Rays per pipeline chunk.  Every chunk pays the tails of its passes once, so bigger is better (8192² in one chunk instead
// of four: 581 -> 565 ms) and 288 GB of HBM can afford it: up to 2^26 rays (14.3 GB of workspace at 213 B/ray; 25.0 GB at the 373 B/ray of a call that asks for end states), halved
// until the workspace fits into a quarter of the memory that is free when it has to be (re)allocated.
template <class R>
uint64_t pick_chunk(const DeviceCtx& d, const StreamState& ss, uint64_t n, bool with_state) {
    const uint64_t cap = d.knobs.chunk > 0 ? (uint64_t)(d.knobs.chunk < 64 ? 64 : d.knobs.chunk) : (1ull << 26);
    uint64_t chunk = n < cap ? n : cap;
    if (workspace_bytes<R>(chunk, with_state) <= ss.ws_bytes) return chunk;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return chunk;
    const size_t budget = free_b / 4;
    while (chunk > (1ull << 20) && workspace_bytes<R>(chunk, with_state) > budget) chunk = (chunk + 1) / 2;
    return chunk;
}
template uint64_t pick_chunk<double>(const DeviceCtx&, const StreamState&, uint64_t, bool);
template uint64_t pick_chunk<float>(const DeviceCtx&, const StreamState&, uint64_t, bool);

int ensure_workspace(DeviceCtx& d, StreamState& ss, size_t bytes, hipStream_t st) {
    if (bytes <= ss.ws_bytes) return RTGR_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(RTGR_ERR_BAD_ARG, "the stream's workspace must grow but the stream is being captured: call "
                                      "rtgr_reserve_workspace for this stream before hipStreamBeginCapture");
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    // the old buffer may be referenced by kernels still in flight on this stream or by a graph captured earlier: retire it
    if (ss.ws) ss.retired.push_back(ss.ws);
    ss.ws = p;
    ss.ws_bytes = bytes;
    return RTGR_OK;
}

int stream_state(DeviceCtx& d, hipStream_t st, StreamState** out) {
    auto it = d.streams.find(st);
    if (it == d.streams.end()) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(RTGR_ERR_BAD_ARG, "first use of this stream while it is being captured: call "
                                          "rtgr_reserve_workspace for this stream before hipStreamBeginCapture");
        StreamState ss;
        // (not zeroed here: reset_kernel zeroes the heads on the launch stream at the start of every chunk.  A hipMemset
        //  would run on the NULL stream, which does not order with a non-blocking caller stream: it landed in the middle
        //  of the first pipeline of a new stream and wiped the early-list cursor — found by the two-streams test)
        HIP_TRY(hipMalloc((void**)&ss.queue, 8 * sizeof(unsigned long long)));
        it = d.streams.emplace(st, ss).first;
    }
    *out = &it->second;
    return RTGR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// contexts
// ---------------------------------------------------------------------------------------------------------------------
void staging_delete(Staging* s) {
    if (!s) return;
    for (auto& b : s->pin_in) b.release();
    for (auto& b : s->pin_out) b.release();
    s->pin_small.release();
    s->d_in.release(); s->d_out.release(); s->d_small.release(); s->d_recv.release();
    for (auto& e : s->ev_in) if (e) (void)hipEventDestroy(e);
    if (s->s_up) (void)hipStreamDestroy(s->s_up);
    if (s->s_comp) (void)hipStreamDestroy(s->s_comp);
    if (s->s_down) (void)hipStreamDestroy(s->s_down);
    delete s;
}

std::mutex g_default_mu;
rtgr_context* g_default = nullptr;
const std::string& last_error_string() { return g_err; }

int staging_of(DeviceCtx& d, Staging** out, int slot) {
    auto& mine = slot ? d.staging2 : d.staging;
    if (!mine) {
        std::unique_ptr<Staging, void (*)(Staging*)> s(new Staging, staging_delete);
        HIP_TRY(hipStreamCreateWithFlags(&s->s_up, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s->s_comp, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s->s_down, hipStreamNonBlocking));
        for (auto& e : s->ev_in) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        mine = std::move(s);
    }
    *out = mine.get();
    return RTGR_OK;
}

void free_device_state(DeviceCtx& d, bool all) {
    DeviceGuard g(d.dev);
    (void)hipDeviceSynchronize();
    for (auto& kv : d.streams) {
        for (void* p : kv.second.retired) (void)hipFree(p);
        kv.second.retired.clear();
        if (all) {
            if (kv.second.ws) (void)hipFree(kv.second.ws);
            if (kv.second.queue) (void)hipFree(kv.second.queue);
        }
    }
    for (auto& kv : d.object_tables) (void)hipFree(kv.second.dev);   // (the device is idle: synchronised above)
    d.object_tables.clear();
    d.object_table_bytes = 0;
    if (all) { d.staging.reset(); d.staging2.reset(); }   // (rtgr_trim releases the staging BUFFERS separately, under the staging's own mutex)
    if (all) {
        d.streams.clear();
        for (auto& m : d.modules) if (m.module && m.owned) (void)hipModuleUnload(m.module);
        d.modules.clear();
        for (auto& t : d.timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
        d.timed.clear();
        for (auto e : d.event_pool) (void)hipEventDestroy(e);
        d.event_pool.clear();
    }
}

int create_context(const int* ids, int n, rtgr_context** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RTGR_ERR_NO_DEVICE, "no HIP device visible; librtgr_hip has no CPU fallback");
    int cur = 0;
    if (!ids) { HIP_TRY(hipGetDevice(&cur)); ids = &cur; n = 1; }
    if (n <= 0 || n > RTGR_MAX_DEVICES) return fail(RTGR_ERR_BAD_ARG, "need 1..RTGR_MAX_DEVICES devices");
    std::unique_ptr<rtgr_context> c(new rtgr_context);
    const Knobs k = knobs_from_env();
    for (int i = 0; i < n; i++) {
        if (ids[i] < 0 || ids[i] >= ndev) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
        hipDeviceProp_t p;
        HIP_TRY(hipGetDeviceProperties(&p, ids[i]));
        if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
            return fail(RTGR_ERR_NO_DEVICE, std::string("librtgr_hip is built for gfx950 only; device is ") + p.gcnArchName);
        std::unique_ptr<DeviceCtx> d(new DeviceCtx);
        d->dev = ids[i];
        d->num_cu = p.multiProcessorCount;
        d->name = std::string(p.name) + " (" + p.gcnArchName + ")";
        d->knobs = k;
        c->devs.push_back(std::move(d));
    }
    // peer access device 0 <-> every other physical device (the gather of rtgr_trace_sharded_device_*).  A failure is not
    // fatal for the context — it is RECORDED per device with its reason, and the gather then stages that device's rows
    // through pinned host memory (or returns RTGR_ERR_HIP naming the pair when the option peer = 1 insists on peer copies).
    c->peer_ok.assign((size_t)n, 1);
    c->peer_why.assign((size_t)n, std::string());
    for (int i = 1; i < n; i++) {
        const int a = c->devs[0]->dev, b = c->devs[i]->dev;
        if (a == b) continue;
        auto enable = [&](int from, int to) -> bool {
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, from, to);
            if (e != hipSuccess || !can) {
                c->peer_why[i] = "hipDeviceCanAccessPeer(" + std::to_string(from) + " -> " + std::to_string(to) + "): " +
                                 (e != hipSuccess ? hipGetErrorString(e) : "no peer access between these devices");
                return false;
            }
            DeviceGuard g(from);
            e = hipDeviceEnablePeerAccess(to, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                c->peer_why[i] = "hipDeviceEnablePeerAccess(" + std::to_string(from) + " -> " + std::to_string(to) + "): " + hipGetErrorString(e);
                return false;
            }
            return true;
        };
        if (!enable(a, b) || !enable(b, a)) c->peer_ok[i] = 0;
        (void)hipGetLastError();  // "already enabled" leaves a sticky error behind
    }
    *out = c.release();
    return RTGR_OK;
}

void destroy_context(rtgr_context* c) {
    if (!c) return;
    for (auto& d : c->devs) free_device_state(*d, true);
    delete c;
}

// ctx == NULL: the process's default context (created on the calling thread's current device on first use)
int resolve_ctx(rtgr_context* in, rtgr_context** out) {
    (void)hipGetLastError();  // every entry point starts here: drop a stale error left by an earlier (or foreign) call
    if (in) { *out = in; return RTGR_OK; }
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        int rc = create_context(nullptr, 0, &g_default);
        if (rc) return rc;
    }
    *out = g_default;
    return RTGR_OK;
}

// the DeviceCtx that owns a device pointer (first entry of the context with that ordinal); NULL pointer: device 0
int device_of(rtgr_context* c, const void* d_ptr, DeviceCtx** out) {
    if (!d_ptr) { *out = c->devs[0].get(); return RTGR_OK; }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, d_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RTGR_ERR_BAD_ARG, "not a device pointer (hipPointerGetAttributes failed)");
    }
    for (auto& d : c->devs)
        if (d->dev == at.device) { *out = d.get(); return RTGR_OK; }
    return fail(RTGR_ERR_BAD_ARG, "the device that owns this pointer is not part of the context");
}

// ---------------------------------------------------------------------------------------------------------------------
// argument conversion
// ---------------------------------------------------------------------------------------------------------------------
// The band of s whose correctly rounded square root (in R) EQUALS r: lo = min{s : sqrt(s) >= r}, hi = min{s : sqrt(s) > r}.
// disk_sign_distance (rtgr_physics.hpp) reads sign(r − RN(sqrt(s))) off these two thresholds, exactly as the IEEE square root
// of obj_distance would give it.  A few nextafter steps around r² (the root maps 1–3 neighbouring s onto one value).
template <class R>
void disk_sqrt_band(R r, R& lo, R& hi) {
    const R inf = std::numeric_limits<R>::infinity();
    if (!(r >= R(0))) { lo = hi = R(0); return; }      // negative (or NaN) radius: sqrt(s) > r for every s >= 0
    if (r == inf) { lo = hi = inf; return; }
    R c = r * r;
    if (!(c < inf)) c = std::numeric_limits<R>::max();
    for (int it = 0; it < 4096 && c > R(0) && std::sqrt(c) >= r; it++) c = std::nextafter(c, -inf);
    for (int it = 0; it < 4096 && std::sqrt(c) < r; it++) c = std::nextafter(c, inf);
    lo = c;
    for (int it = 0; it < 4096 && std::sqrt(c) <= r; it++) c = std::nextafter(c, inf);
    hi = c;
}

// The (metric enum | generic flag, spin) a scene selects among the kernels' instantiations — what dispatch() below switches on, and
// what a run-time unit without a metric of its own is built for (rtgr_user_unit_desc, rtgr_user_unit.hip.in).
void scene_variant(const rtgr_scene* s, uint32_t* metric, bool* spin) {
    const uint32_t kind = s->metric & ~RTGR_METRIC_GENERIC;
    const bool generic = (s->metric & RTGR_METRIC_GENERIC) != 0 && kind != RTGR_MINKOWSKI;
    *metric = kind | (generic ? RTGR_METRIC_GENERIC : 0u);
    *spin = kind == RTGR_MINKOWSKI ? false : (generic ? true : s->a != 0.0);
}

// The load-time probe of a unit of OBJECTS traces a scene of built-in objects (it knows no parameters of the user's) and must still run
// the UNIT's kernels, not the library's: while this is set on the calling thread, a scene that names a unit runs with it even though
// nothing in the scene requires one.  (Everywhere else a built-in scene ignores rtgr_scene.user_metric, as it always has.)
thread_local bool tl_probe_forces_unit = false;
// … and the probe (and rtgr_scene_check) choose the launch options of THEIR calls — pass structure, queue order, grid size — without
// touching the device's options, which other threads' calls on the same device read: null = the device's options decide.
thread_local const Knobs* tl_knobs_override = nullptr;

// one object of the caller's list -> its device form (the scalar type's values; a disk's sign thresholds)
template <class R>
static int convert_object(const rtgr_object& in, DevObject<R>& d) {
    if (in.kind < RTGR_PLANE || in.kind > RTGR_USER_OBJECT)
        return fail(RTGR_ERR_BAD_ARG, "unknown object kind (abstract Object has no distance)");
    d.kind = in.kind;
    d.type = in.kind == RTGR_USER_OBJECT ? in.type : 0u;
    for (int q = 0; q < 9; q++) d.p[q] = (R)in.p[q];
    if (in.kind == RTGR_DISK) {   // p[3..6]: the scan's sign thresholds on x² + y² (device-side only; the ABI's disk is p[0..2])
        disk_sqrt_band<R>(d.p[1], d.p[3], d.p[4]);
        disk_sqrt_band<R>(d.p[2], d.p[5], d.p[6]);
    }
    return RTGR_OK;
}

// The device table of the objects beyond the argument block (ObjectTable, rtgr_host.hpp): found by content, uploaded on first sight.
static int object_table(DeviceCtx& D, const std::vector<char>& content, hipStream_t st, const void** dev) {
    const uint64_t key = fnv1a(content);
    auto range = D.object_tables.equal_range(key);
    for (auto it = range.first; it != range.second; ++it)
        if (it->second.content == content) { *dev = it->second.dev; return RTGR_OK; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(RTGR_ERR_BAD_ARG, "a scene of more than RTGR_MAX_OBJECTS objects is seen for the first time while the stream is being captured "
                                      "(its object table must be uploaded): trace the scene once before hipStreamBeginCapture");
    if (D.object_tables.size() >= OBJECT_TABLES_MAX || D.object_table_bytes + content.size() > OBJECT_TABLES_BYTES) {
        // a caller that animates a long list: start over (nothing in flight may read a freed table)
        HIP_TRY(hipDeviceSynchronize());
        for (auto& kv : D.object_tables) (void)hipFree(kv.second.dev);
        D.object_tables.clear();
        D.object_table_bytes = 0;
    }
    ObjectTable t;
    HIP_TRY(hipMalloc(&t.dev, content.size()));
    const hipError_t e = hipMemcpy(t.dev, content.data(), content.size(), hipMemcpyHostToDevice);   // blocking: complete before any launch
    if (e != hipSuccess) { (void)hipFree(t.dev); return fail(RTGR_ERR_HIP, std::string("hipMemcpy(object table): ") + hipGetErrorString(e)); }
    t.content = content;
    D.object_table_bytes += content.size();
    *dev = t.dev;
    D.object_tables.emplace(key, std::move(t));
    return RTGR_OK;
}

// ---- groups of a long list's spheres (DevScene, rtgr_args.hpp: GROUPS) ------------------------------------------------------------
struct SphereGroup { uint32_t first, count; };   // positions in the device list
// `order` holds the spheres in the caller's order; on return: [0, *nloose) the spheres that join no group, then the groups' members,
// group by group (`groups`: their positions).  Groups are the leaves of median splits of the centres along the widest axis — a k-d
// tree's leaves, <= RTGR_GROUP_MAX members each (half that for lists of fewer than 36 spheres); a sphere much larger than the list's typical one (a sky sphere around the scene) would
// make its group's bounding sphere as large as itself and stays loose.  Lists with fewer than two full groups, or with a non-finite
// centre or radius among the spheres, get no groups.  Deterministic: ties are broken by the caller's index.
// `supers`: the second level — runs of neighbouring groups (first / count are GROUP indices): the nodes of the same split tree that hold
// at most 8 leaves' worth of spheres; only lists of RTGR_SUPER_FROM groups and more get them.
constexpr size_t RTGR_SUPER_FROM = 24;
static void group_spheres(const rtgr_object* objs, std::vector<uint32_t>& order, uint32_t* nloose, std::vector<SphereGroup>& groups,
                          std::vector<SphereGroup>& supers, size_t leaf = 0, double limit = std::numeric_limits<double>::max()) {
    *nloose = 0;
    groups.clear();
    supers.clear();
    const size_t n = order.size();
    if (n < 2 * (size_t)RTGR_GROUP_MAX) return;
    std::vector<double> radii(n);
    for (size_t k = 0; k < n; k++) {
        const rtgr_object& o = objs[order[k]];
        // (finite in the kernels' scalar type too: `limit` is its largest value — a centre that becomes inf there has no bounding sphere)
        if (!(std::fabs(o.p[1]) <= limit) || !(std::fabs(o.p[2]) <= limit) || !(std::fabs(o.p[3]) <= limit) || !(std::fabs(o.p[8]) <= limit)) return;
        radii[k] = std::fabs(o.p[8]);
    }
    std::vector<double> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const double big = 4.0 * sorted[n / 2];
    std::vector<uint32_t> loose, rest;
    // (… and an inside-out sphere, R < 0: the resolve kernel's bound for a group's members — select_objects — wants R >= 0)
    for (size_t k = 0; k < n; k++) ((radii[k] > big || objs[order[k]].p[8] < 0) ? loose : rest).push_back(order[k]);
    if (rest.size() < 2 * (size_t)RTGR_GROUP_MAX) return;
    // (leaf = 0: automatic — a few dozen spheres spread over a scene make better groups of <= 4: 2048², 24 objects 35.5 -> 33.8 ms, 32:
    //  42.4 -> 41.4, but 40: 43.9 -> 45.7; profiles/r06/groups_small_lists.log)
    if (leaf == 0) leaf = rest.size() < 36 ? RTGR_GROUP_MAX / 2 : RTGR_GROUP_MAX;
    std::vector<std::pair<size_t, size_t>> todo{{0, rest.size()}}, leaves;
    while (!todo.empty()) {
        const auto [lo, hi] = todo.back();
        todo.pop_back();
        if (hi - lo <= leaf) { leaves.push_back({lo, hi}); continue; }
        double mn[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, mx[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (size_t k = lo; k < hi; k++)
            for (int q = 0; q < 3; q++) { mn[q] = std::min(mn[q], objs[rest[k]].p[1 + q]); mx[q] = std::max(mx[q], objs[rest[k]].p[1 + q]); }
        int ax = 0;
        for (int q = 1; q < 3; q++) if (mx[q] - mn[q] > mx[ax] - mn[ax]) ax = q;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(rest.begin() + lo, rest.begin() + mid, rest.begin() + hi, [&](uint32_t a, uint32_t b) {
            const double ca = objs[a].p[1 + ax], cb = objs[b].p[1 + ax];
            return ca < cb || (ca == cb && a < b);
        });
        todo.push_back({mid, hi});
        todo.push_back({lo, mid});
    }
    std::sort(leaves.begin(), leaves.end());
    for (auto& lf : leaves) std::sort(rest.begin() + lf.first, rest.begin() + lf.second);   // (members in the caller's order)
    *nloose = (uint32_t)loose.size();
    order = loose;
    order.insert(order.end(), rest.begin(), rest.end());
    for (auto& lf : leaves) groups.push_back({(uint32_t)(loose.size() + lf.first), (uint32_t)(lf.second - lf.first)});
    if (leaves.size() < RTGR_SUPER_FROM) return;
    // the second level: the same tree (the splits above cut every range at lo + (hi - lo) / 2), stopped at nodes of <= 8 leaves' worth —
    // more for very long lists: with G groups in runs of r, a step asks G / r runs + r groups of every run it cannot rule out (a few),
    // least for r ~ sqrt(G / 3) (8 up to ~1500 spheres; 70 for 120000)
    const size_t per_run = std::max<size_t>(8, (size_t)std::sqrt((double)leaves.size() / 3.0));
    std::vector<std::pair<size_t, size_t>> nodes;
    todo.assign(1, {0, rest.size()});
    while (!todo.empty()) {
        const auto [lo, hi] = todo.back();
        todo.pop_back();
        if (hi - lo <= per_run * leaf) { nodes.push_back({lo, hi}); continue; }
        const size_t mid = lo + (hi - lo) / 2;
        todo.push_back({mid, hi});
        todo.push_back({lo, mid});
    }
    std::sort(nodes.begin(), nodes.end());
    size_t g = 0;
    for (auto& nd : nodes) {   // (leaves are sorted by position and nest in the nodes)
        const size_t g0 = g;
        while (g < leaves.size() && leaves[g].first < nd.second) g++;
        supers.push_back({(uint32_t)g0, (uint32_t)(g - g0)});
    }
}
// … and a group's bounding sphere from its members AS THE KERNELS SEE THEM (the scalar type's values): centre = the middle of the
// centres' box, radius = max (|c_i − C| + |r_i|), evaluated in double and rounded UP into R with a margin of 64 ulp — what the reach
// test's argument needs is containment, not tightness.
template <class R>
static void bounding_sphere(const DevObject<R>* t, const SphereGroup& g, DevObject<R>& out) {
    double mn[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, mx[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (uint32_t k = g.first; k < g.first + g.count; k++)
        for (int q = 0; q < 3; q++) { mn[q] = std::min(mn[q], (double)t[k].p[1 + q]); mx[q] = std::max(mx[q], (double)t[k].p[1 + q]); }
    std::memset(&out, 0, sizeof out);
    out.kind = RTGR_SPHERE;
    out.type = g.first;
    out.orig = g.count;
    for (int q = 0; q < 3; q++) out.p[1 + q] = (R)(0.5 * mn[q] + 0.5 * mx[q]);
    double rad = 0;
    for (uint32_t k = g.first; k < g.first + g.count; k++) {
        double d2 = 0;
        for (int q = 0; q < 3; q++) { const double e = (double)t[k].p[1 + q] - (double)out.p[1 + q]; d2 += e * e; }
        rad = std::max(rad, std::sqrt(d2) + std::fabs((double)t[k].p[8]));
    }
    rad = rad * (1.0 + 64.0 * (double)std::numeric_limits<R>::epsilon()) + (double)std::numeric_limits<R>::min();
    R r = (R)rad;
    if ((double)r < rad) r = std::nextafter(r, std::numeric_limits<R>::infinity());
    out.p[8] = r;
}
// … and a run of groups' bounding sphere: over ALL the members of its groups (they are neighbours in the device list too)
template <class R>
static void super_sphere(const DevObject<R>* t, const std::vector<SphereGroup>& groups, const SphereGroup& run, DevObject<R>& out) {
    const SphereGroup& a = groups[run.first];
    const SphereGroup& b = groups[run.first + run.count - 1];
    bounding_sphere<R>(t, SphereGroup{a.first, b.first + b.count - a.first}, out);
    out.type = run.first;
    out.orig = run.count;
}

template <class R>
int convert_scene(DeviceCtx& D, const rtgr_scene* s, DevScene<R>& d, const UserModule** user, hipStream_t st) {
    if (!s) return fail(RTGR_ERR_BAD_ARG, "scene is NULL");
    if ((s->metric & ~RTGR_METRIC_GENERIC) > RTGR_USER) return fail(RTGR_ERR_BAD_ARG, "unknown metric enum");
    *user = nullptr;
    if (s->nobj > RTGR_MAX_OBJECTS && !s->objects)
        return fail(RTGR_ERR_BAD_ARG, "more than RTGR_MAX_OBJECTS objects: hand the list over through rtgr_scene.objects (any length)");
    if (s->nobj > RTGR_OBJECTS_LIMIT) return fail(RTGR_ERR_BAD_ARG, "more than RTGR_OBJECTS_LIMIT objects");
    const rtgr_object* objs = scene_objects(s);
    const bool user_metric = (s->metric & ~RTGR_METRIC_GENERIC) == RTGR_USER;
    bool user_objects = tl_probe_forces_unit && s->user_metric != 0 && !user_metric;
    for (uint32_t o = 0; o < s->nobj; o++) user_objects = user_objects || objs[o].kind == RTGR_USER_OBJECT;
    if (user_metric || user_objects) {
        const char* what = user_metric ? "RTGR_USER" : "RTGR_USER_OBJECT";
        if (D.modules.empty())
            return fail(RTGR_ERR_BAD_ARG, std::string(what) + ": no run-time unit loaded (rtgr_user_metric_load / rtgr_user_unit_compile)");
        *user = D.find_module(s->user_metric);
        if (!*user)
            return fail(RTGR_ERR_BAD_ARG, std::string(what) + ": rtgr_scene.user_metric names a unit that is not loaded in this "
                                          "context (a scene only ever runs with the kernels of its own unit)");
        const UserModule& U = **user;
        if (user_metric && !U.has_metric)
            return fail(RTGR_ERR_BAD_ARG, "RTGR_USER: this unit defines no metric (it was built for a built-in one: rtgr_user_unit_info)");
        if (user_objects && !U.has_objects)
            return fail(RTGR_ERR_BAD_ARG, "RTGR_USER_OBJECT: this unit's source defines no rtgr_user_distance / rtgr_user_objcolor");
        if (!user_metric) {   // the unit's kernels are ONE built-in metric variant's: the scene's must be that one
            uint32_t mv; bool sp;
            scene_variant(s, &mv, &sp);
            if (U.has_metric || U.metric != mv || U.spin != sp)
                return fail(RTGR_ERR_BAD_ARG, "RTGR_USER_OBJECT: this unit's kernels were built for another metric variant (metric enum / "
                                              "RTGR_METRIC_GENERIC / a != 0 differ from the scene's): build one for THIS scene — "
                                              "rtgr_user_unit_compile(ctx, source, stationary, &scene, &id)");
        }
        if (sizeof(R) != 8 && !U.full10_f32)
            return fail(RTGR_ERR_BAD_ARG, "this unit carries no Float32 kernels");
    }
    std::memset(&d, 0, sizeof d);
    d.metric = s->metric & ~RTGR_METRIC_GENERIC;
    d.nobj = s->nobj;
    d.M = (R)s->M;
    d.a = (R)s->a;
    int rc;
    const uint32_t n0 = s->nobj < (uint32_t)RTGR_MAX_OBJECTS ? s->nobj : (uint32_t)RTGR_MAX_OBJECTS;
    // the device list is REGROUPED (DevScene, rtgr_args.hpp): spheres first, then the rest, both in the caller's order, every object
    // with its original index — the integrate kernels walk the spheres without a dispatch on the kind; the colour rule breaks ties
    // by the original index and reports it, so no result depends on the regrouping
    std::vector<uint32_t> order;
    order.reserve(s->nobj);
    for (uint32_t o = 0; o < s->nobj; o++) if (objs[o].kind == RTGR_SPHERE) order.push_back(o);
    d.nsph = (uint32_t)order.size();
    // … and the spheres of a LONG list in groups of neighbours (DevScene: GROUPS)
    std::vector<SphereGroup> groups, supers;
    const long kg = tl_knobs_override ? tl_knobs_override->groups : D.knobs.groups;   // (0: off; 1: on; >= 2: on, with that many spheres per group at most — experiments)
    if (s->nobj > n0 && kg)
        group_spheres(objs, order, &d.nloose, groups, supers, kg >= 2 ? (size_t)kg : (size_t)0, (double)std::numeric_limits<R>::max());
    for (uint32_t o = 0; o < s->nobj; o++) if (objs[o].kind != RTGR_SPHERE) order.push_back(o);
    for (uint32_t k = 0; k < n0; k++) {
        if ((rc = convert_object<R>(objs[order[k]], d.obj[k]))) return rc;
        d.obj[k].orig = order[k];
    }
    if (s->nobj > n0) {   // a long list: a device table of ALL of it (+ its groups), shared by every call with the same list
        std::vector<char> content((size_t)(s->nobj + groups.size() + supers.size()) * sizeof(DevObject<R>), 0);
        DevObject<R>* t = (DevObject<R>*)content.data();
        for (uint32_t k = 0; k < s->nobj; k++) {
            if ((rc = convert_object<R>(objs[order[k]], t[k]))) return rc;
            t[k].orig = order[k];
        }
        for (size_t g = 0; g < groups.size(); g++) bounding_sphere<R>(t, groups[g], t[s->nobj + g]);
        for (size_t u = 0; u < supers.size(); u++) super_sphere<R>(t, groups, supers[u], t[s->nobj + groups.size() + u]);
        d.ngroups = (uint32_t)groups.size();
        d.nsuper = (uint32_t)supers.size();
        const void* dev = nullptr;
        if ((rc = object_table(D, content, st, &dev))) return rc;
        d.more = (const DevObject<R>*)dev + n0;
    }
    return RTGR_OK;
}
template <class R>
int convert_solver(const rtgr_solver* s, DevSolver<R>& d) {
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
void convert_camera(const rtgr_camera* c, DevCamera<R>& d) {
    for (int a = 0; a < 4; a++) {
        d.pos[a] = (R)c->pos[a];
        d.widthx[a] = (R)c->widthx[a];
        d.widthy[a] = (R)c->widthy[a];
        d.normal[a] = (R)c->normal[a];
    }
}

int dispatch(LaunchEnv& E, const TraceArgs<double>& A, bool generic, bool spin, hipStream_t st) {
    if (generic || E.user) return launch_f64_generic(E, A, st);   // (a scene with a run-time unit launches the unit's kernels from there)
    switch (A.sc.metric) {
        case RTGR_MINKOWSKI: return launch_f64_mink(E, A, st);
        case RTGR_KS_REF: return launch_f64_ksref(E, A, spin, st);
        default: return launch_f64_kstrue(E, A, spin, st);
    }
}
int dispatch(LaunchEnv& E, const TraceArgs<float>& A, bool generic, bool spin, hipStream_t st) {
    if (generic || E.user) return launch_f32_generic(E, A, st);
    return launch_f32_closed(E, A, spin, st);
}


// Enqueue the pipeline for rows of a canvas on device D, stream st.  The caller holds no lock; this takes D.mu for the
// duration of the enqueue.
template <class R>
int trace_device(DeviceCtx& D, const rtgr_scene* scene, const rtgr_solver* opt, const R* d_state0, const rtgr_camera* cam,
                 uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, R* d_rgb, const rtgr_ray_outputs* out,
                 rtgr_counters* d_counters, hipStream_t st, uint64_t jstride, uint64_t nrows_strided, const Window* win) {
    DeviceGuard guard(D.dev);
    if (!guard.ok) return fail(RTGR_ERR_HIP, "hipSetDevice failed");
    int rc;
    // a scene whose user objects bring their own reach bound is checked the first time it is seen (blocking, a few ms; takes D.mu itself)
    if ((rc = auto_scene_check<R>(D, scene, opt, d_state0, cam, ni, nj, j0, j1, jstride, nrows_strided, st))) return rc;
    std::lock_guard<std::mutex> lk(D.mu);
    TraceArgs<R> A;
    std::memset(&A, 0, sizeof A);
    const UserModule* user = nullptr;
    if ((rc = convert_scene<R>(D, scene, A.sc, &user, st))) return rc;
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
        A.hit32 = out->hit32;
        if (A.hit && A.sc.nobj > 255u)
            return fail(RTGR_ERR_BAD_ARG, "rtgr_ray_outputs.hit is a byte per ray and the scene has more than 255 objects: ask for hit32");
        A.n_accept = out->n_accept;
        A.n_reject = out->n_reject;
        if (out->redshift) {
            if (A.sc.metric == RTGR_USER && !(sizeof(R) == 8 ? user->redshift : user->redshift_f32))
                return fail(RTGR_ERR_BAD_ARG, "rtgr_ray_outputs.redshift: this user-metric code object carries no rtgr_user_redshift kernel (rebuild the unit)");
            if (!out->state_end || !(out->hit || out->hit32))
                return fail(RTGR_ERR_BAD_ARG, "rtgr_ray_outputs.redshift needs state_end and hit (or hit32) in the same call");
        }
    }
    if (win) { A.plane_stride = win->plane_stride; A.out_offset = win->out_offset; A.nan_flag = win->nan_flag; }
    A.counters = (unsigned long long*)d_counters;
    const bool spin = scene->a != 0.0;
    const bool generic = ((scene->metric & RTGR_METRIC_GENERIC) != 0 && A.sc.metric != RTGR_MINKOWSKI) || A.sc.metric == RTGR_USER;
    if ((tl_knobs_override ? tl_knobs_override->tile : D.knobs.tile)) {   // (the probe and the scene check bring their own options: tile = 0)
        if (generic || user) return fail(RTGR_ERR_BAD_ARG, "RTGR_METRIC_GENERIC and run-time units need the persistent pipeline (option tile = 0)");
        if (win && (win->plane_stride || win->out_offset)) return fail(RTGR_ERR_BAD_ARG, "the tile kernel writes whole slabs only");
    }
    StreamState* ss = nullptr;
    if ((rc = stream_state(D, st, &ss))) return rc;
    LaunchEnv E{D, *ss, user, win ? win->after_setup : nullptr, tl_knobs_override};
    rc = dispatch(E, A, generic, spin, st);
    if (rc) return rc;
    if (out && out->redshift) {   // one more kernel behind the pipeline: needs the end states and the hit map it wrote
        const uint64_t nr = ni * nrows;
        if (A.sc.metric == RTGR_USER) {
            HIP_TRY(launch_module(sizeof(R) == 8 ? user->redshift : user->redshift_f32, (unsigned)((nr + 255) / 256), 256, st, A.sc, A.cam,
                                  A.state0, ni, nj, j0, jstride, nr, A.out_offset, (const R*)A.state_end, (const uint8_t*)A.hit,
                                  (const uint32_t*)A.hit32, (R*)out->redshift));
        } else if constexpr (sizeof(R) == 8) {
            rc = misc_redshift_f64(A.sc, A.cam, (const double*)A.state0, ni, nj, j0, jstride, nr, A.out_offset,
                                   (const double*)A.state_end, A.hit, A.hit32, (double*)out->redshift, st);
        } else {
            rc = misc_redshift_f32(A.sc, A.cam, (const float*)A.state0, ni, nj, j0, jstride, nr, A.out_offset,
                                   (const float*)A.state_end, A.hit, A.hit32, (float*)out->redshift, st);
        }
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    return RTGR_OK;
}

template int convert_scene<double>(DeviceCtx&, const rtgr_scene*, DevScene<double>&, const UserModule**, hipStream_t);
template int convert_scene<float>(DeviceCtx&, const rtgr_scene*, DevScene<float>&, const UserModule**, hipStream_t);
template int convert_solver<double>(const rtgr_solver*, DevSolver<double>&);
template int convert_solver<float>(const rtgr_solver*, DevSolver<float>&);
template void convert_camera<double>(const rtgr_camera*, DevCamera<double>&);
template void convert_camera<float>(const rtgr_camera*, DevCamera<float>&);
template int trace_device<double>(DeviceCtx&, const rtgr_scene*, const rtgr_solver*, const double*, const rtgr_camera*, uint64_t, uint64_t, uint64_t,
                                  uint64_t, double*, const rtgr_ray_outputs*, rtgr_counters*, hipStream_t, uint64_t, uint64_t, const Window*);
template int trace_device<float>(DeviceCtx&, const rtgr_scene*, const rtgr_solver*, const float*, const rtgr_camera*, uint64_t, uint64_t, uint64_t,
                                 uint64_t, float*, const rtgr_ray_outputs*, rtgr_counters*, hipStream_t, uint64_t, uint64_t, const Window*);

// completed timed launches -> the device's accumulators ([0..3] pipeline kernels, [4..5] the multi-device exchange); D.mu held
int collect_timed(DeviceCtx& d) {
    for (auto& t : d.timed) {
        HIP_TRY(hipEventSynchronize(t.b));
        float e = 0.f;
        HIP_TRY(hipEventElapsedTime(&e, t.a, t.b));
        d.acc_ms[t.which] += e;
        d.acc_n[t.which] += 1;
        d.event_pool.push_back(t.a);
        d.event_pool.push_back(t.b);
    }
    d.timed.clear();
    return RTGR_OK;
}

uint64_t fnv1a(const std::vector<char>& b) {
    uint64_t h = 1469598103934665603ull;
    for (char ch : b) { h ^= (unsigned char)ch; h *= 1099511628211ull; }
    return h ? h : 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// entry points: lifecycle, options, timing, the device-resident hot path
// ---------------------------------------------------------------------------------------------------------------------
int api::abi_version(void) { return RTGR_ABI_VERSION; }
const char* api::last_error(void) { return g_err.c_str(); }

int api::create(const int* device_ids, int n_devices, rtgr_context** ctx_out) {
    if (!ctx_out) return fail(RTGR_ERR_BAD_ARG, "ctx_out is NULL");
    *ctx_out = nullptr;
    (void)hipGetLastError();
    return create_context(device_ids, n_devices, ctx_out);
}
int api::destroy(rtgr_context* ctx) {
    if (!ctx) return RTGR_OK;
    { std::lock_guard<std::mutex> lk(g_default_mu); if (ctx == g_default) g_default = nullptr; }
    destroy_context(ctx);
    return RTGR_OK;
}
int api::context_devices(rtgr_context* ctx) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    return rc ? rc : (int)c->devs.size();
}
int api::trim(rtgr_context* ctx) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    for (auto& d : c->devs) {
        // Host-pointer calls hold Staging::mu for their whole duration and take D.mu inside it (to enqueue), so the staging
        // is trimmed FIRST and under its own mutex only (a call in flight finishes first, the next one re-allocates); the
        // struct itself — mutex, streams, events — goes with the context, never here.
        for (int slot = 0; slot < 2; slot++) {
            Staging* s = nullptr;
            { std::lock_guard<std::mutex> lk(d->mu); s = slot ? d->staging2.get() : d->staging.get(); }
            if (!s) continue;
            DeviceGuard g(d->dev);
            std::lock_guard<std::mutex> ls(s->mu);
            (void)hipStreamSynchronize(s->s_up); (void)hipStreamSynchronize(s->s_comp); (void)hipStreamSynchronize(s->s_down);
            for (auto& b : s->pin_in) b.release();
            for (auto& b : s->pin_out) b.release();
            s->pin_small.release();
            s->d_in.release(); s->d_out.release(); s->d_small.release(); s->d_recv.release();
        }
        std::lock_guard<std::mutex> lk(d->mu);
        free_device_state(*d, false);
    }
    return RTGR_OK;
}
int api::init(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(RTGR_ERR_NO_DEVICE, "no HIP device visible; librtgr_hip has no CPU fallback");
    if (device >= n) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    int dev = device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    else HIP_TRY(hipSetDevice(dev));
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default && g_default->devs.size() == 1 && g_default->devs[0]->dev == dev) return RTGR_OK;
    if (g_default) { destroy_context(g_default); g_default = nullptr; }
    return create_context(&dev, 1, &g_default);
}
int api::shutdown(void) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (g_default) { destroy_context(g_default); g_default = nullptr; }
    return RTGR_OK;
}
int api::solver_defaults(rtgr_solver* s, int is_f32) {
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
int api::device_info(rtgr_context* ctx, int index, char* name, uint64_t name_len, int* n_cu, int* clock_mhz, int* wavefront) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (index < 0 || index >= (int)c->devs.size()) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, c->devs[index]->dev));
    if (name && name_len) std::snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (wavefront) *wavefront = p.warpSize;
    return RTGR_OK;
}

int api::set_option(rtgr_context* ctx, const char* name, long value) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    for (auto& d : c->devs) {
        std::lock_guard<std::mutex> lk(d->mu);
        long* s = knob_slot(d->knobs, name);
        if (!s) return fail(RTGR_ERR_BAD_ARG, std::string("unknown option ") + (name ? name : "(null)"));
        *s = value;
    }
    return RTGR_OK;
}
int api::get_option(rtgr_context* ctx, const char* name, long* value) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!value) return fail(RTGR_ERR_BAD_ARG, "value is NULL");
    long* s = knob_slot(c->devs[0]->knobs, name);
    if (!s) return fail(RTGR_ERR_BAD_ARG, std::string("unknown option ") + (name ? name : "(null)"));
    *value = *s;
    return RTGR_OK;
}

int api::timing_enable(rtgr_context* ctx, int index, int on) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (index < 0 || index >= (int)c->devs.size()) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    std::lock_guard<std::mutex> lk(c->devs[index]->mu);
    c->devs[index]->timing = on != 0;
    return RTGR_OK;
}
int api::timing_read(rtgr_context* ctx, int index, double ms[4], uint64_t launches[4]) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!ms || !launches) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (index < 0 || index >= (int)c->devs.size()) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    DeviceCtx& d = *c->devs[index];
    DeviceGuard guard(d.dev);
    std::lock_guard<std::mutex> lk(d.mu);
    if ((rc = collect_timed(d))) return rc;
    for (int w = 0; w < 4; w++) { ms[w] = d.acc_ms[w]; launches[w] = d.acc_n[w]; d.acc_ms[w] = 0.0; d.acc_n[w] = 0; }
    return RTGR_OK;
}
int api::timing_read_exchange(rtgr_context* ctx, int index, double ms[2], uint64_t launches[2]) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (!ms || !launches) return fail(RTGR_ERR_BAD_ARG, "NULL argument");
    if (index < 0 || index >= (int)c->devs.size()) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    DeviceCtx& d = *c->devs[index];
    DeviceGuard guard(d.dev);
    std::lock_guard<std::mutex> lk(d.mu);
    if ((rc = collect_timed(d))) return rc;
    for (int w = 0; w < 2; w++) { ms[w] = d.acc_ms[4 + w]; launches[w] = d.acc_n[4 + w]; d.acc_ms[4 + w] = 0.0; d.acc_n[4 + w] = 0; }
    return RTGR_OK;
}
int api::peer_access(rtgr_context* ctx, int index, char* why, uint64_t why_len) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    if (index < 0 || index >= (int)c->devs.size()) return fail(RTGR_ERR_BAD_ARG, "device index out of range");
    if (why && why_len) std::snprintf(why, (size_t)why_len, "%s", c->peer_why[(size_t)index].c_str());
    return c->peer_ok[(size_t)index] ? 1 : 0;
}

int api::reserve_workspace(rtgr_context* ctx, const void* d_any, void* stream, uint64_t n_rays, int with_state_end, int is_f32) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(ctx, &c);
    if (rc) return rc;
    DeviceCtx* D = nullptr;
    if ((rc = device_of(c, d_any, &D))) return rc;
    DeviceGuard guard(D->dev);
    std::lock_guard<std::mutex> lk(D->mu);
    StreamState* ss = nullptr;
    if ((rc = stream_state(*D, (hipStream_t)stream, &ss))) return rc;
    const bool ws = with_state_end != 0;
    const size_t bytes = is_f32 ? workspace_bytes<float>(pick_chunk<float>(*D, *ss, n_rays, ws), ws)
                                : workspace_bytes<double>(pick_chunk<double>(*D, *ss, n_rays, ws), ws);
    return ensure_workspace(*D, *ss, bytes, (hipStream_t)stream);
}

int api::trace_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* d_state0,
                          const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* d_rgb,
                          const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    if (!d_rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    RESOLVE_DEVICE(d_rgb);
    return trace_device<double>(*D, scene, opt, d_state0, cam, ni, nj, j0, j1, d_rgb, out, d_counters, (hipStream_t)stream);
}
int api::trace_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* d_state0,
                          const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* d_rgb,
                          const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    if (!d_rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    RESOLVE_DEVICE(d_rgb);
    return trace_device<float>(*D, scene, opt, d_state0, cam, ni, nj, j0, j1, d_rgb, out, d_counters, (hipStream_t)stream);
}
int api::trace_rows_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                               uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, double* d_rgb,
                               const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    if (!cam) return fail(RTGR_ERR_BAD_ARG, "camera is NULL");
    if (!d_rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    RESOLVE_DEVICE(d_rgb);
    return trace_device<double>(*D, scene, opt, nullptr, cam, ni, nj, j0, j0 + 1, d_rgb, out, d_counters, (hipStream_t)stream, jstride, nrows);
}
int api::trace_rows_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                               uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, float* d_rgb,
                               const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream) {
    if (!cam) return fail(RTGR_ERR_BAD_ARG, "camera is NULL");
    if (!d_rgb) return fail(RTGR_ERR_BAD_ARG, "rgb is NULL");
    RESOLVE_DEVICE(d_rgb);
    return trace_device<float>(*D, scene, opt, nullptr, cam, ni, nj, j0, j0 + 1, d_rgb, out, d_counters, (hipStream_t)stream, jstride, nrows);
}

}  // namespace rtgr

// A test hook, not part of include/rtgr.h (tests/test_host_logic.py; host code only — runs without a GPU): how convert_scene lays a
// list's SPHERES out — `order` (n entries: indices into objs, loose spheres first, then the groups' members), the number of loose
// ones, and per group {centre x, y, z, radius, first position, count} as the kernels of the scalar type get them.  objs[] must all be
// RTGR_SPHEREs.  Returns the number of groups (0: the list gets none), or -1 when `cap` rows do not hold them; *nsuper rows of runs of
// groups follow the groups' rows (first / count: group indices).
extern "C" int rtgr_testhook_group_spheres(const rtgr_object* objs, uint32_t n, int is_f32, uint32_t* order_out, uint32_t* nloose,
                                           double* groups_out, uint32_t cap, uint32_t* nsuper) {
    using namespace rtgr;
    std::vector<uint32_t> order(n);
    for (uint32_t k = 0; k < n; k++) order[k] = k;
    std::vector<SphereGroup> groups, supers;
    group_spheres(objs, order, nloose, groups, supers, 0, is_f32 ? (double)std::numeric_limits<float>::max() : std::numeric_limits<double>::max());
    for (uint32_t k = 0; k < n; k++) order_out[k] = order[k];
    if (groups.size() + supers.size() > cap) return -1;
    if (nsuper) *nsuper = (uint32_t)supers.size();
    auto fill = [&](auto zero) {
        typedef decltype(zero) R;
        std::vector<DevObject<R>> t(n);
        for (uint32_t k = 0; k < n; k++) (void)convert_object<R>(objs[order[k]], t[k]);
        for (size_t g = 0; g < groups.size(); g++) {
            DevObject<R> G;
            bounding_sphere<R>(t.data(), groups[g], G);
            double* o = groups_out + 6 * g;
            o[0] = (double)G.p[1]; o[1] = (double)G.p[2]; o[2] = (double)G.p[3]; o[3] = (double)G.p[8]; o[4] = G.type; o[5] = G.orig;
        }
        for (size_t u = 0; u < supers.size(); u++) {   // (the runs of groups follow the groups; first / count are group indices)
            DevObject<R> G;
            super_sphere<R>(t.data(), groups, supers[u], G);
            double* o = groups_out + 6 * (groups.size() + u);
            o[0] = (double)G.p[1]; o[1] = (double)G.p[2]; o[2] = (double)G.p[3]; o[3] = (double)G.p[8]; o[4] = G.type; o[5] = G.orig;
        }
    };
    if (is_f32) fill(0.0f); else fill(0.0);
    return (int)groups.size();
}

#ifdef RTGR_ROOT_STATS
// (debug builds export two symbols that are not part of include/rtgr.h)
using namespace rtgr;
extern "C" {
// debug builds only: a device buffer the NEAR pass writes per-wave {start, end, iterations, rays} and per-ray stays into
// (tools/wave_timeline.py)
int rtgr_debug_set_buffer(void* d_buf) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(nullptr, &c);
    if (rc) return rc;
    c->devs[0]->dbg = (unsigned long long*)d_buf;
    return RTGR_OK;
}
// debug builds only: copy the head of the default stream's workspace (the event records) to the host
int rtgr_debug_workspace(void* stream, void* dst, uint64_t bytes) {
    rtgr_context* c = nullptr;
    int rc = resolve_ctx(nullptr, &c);
    if (rc) return rc;
    DeviceCtx& d = *c->devs[0];
    auto it = d.streams.find((hipStream_t)stream);
    if (it == d.streams.end() || bytes > it->second.ws_bytes) return fail(RTGR_ERR_BAD_ARG, "no workspace / too many bytes");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(dst, it->second.ws, bytes, hipMemcpyDeviceToHost));
    return RTGR_OK;
}
}  // extern "C"
#endif
