/*
 * rtgr.h — C ABI of the MI355X-native geodesic ray tracer (librtgr_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of eschnett/RayTraceGR.jl: the per-pixel
 * geodesic integration + object intersection + colouring that the reference performs in
 *
 *     trace_rays(metric, objs::Vector{Object{T}}, c::Canvas{T})   src/RayTraceGR.jl:482-536
 *
 * i.e. everything between building the EnsembleProblem (:488-509), the
 * `solve(probs, Tsit5(), callback=cb, trajectories=N, reltol=tol, abstol=tol)` call (:510-511) and the
 * colouring loop (:513-533), plus the camera that feeds it (make_canvas, :457-478) and the legacy
 * single-ray shape `trace_ray(metric, objs, cb, p)::Pixel` (test/runtests.jl:65-79).
 *
 * Every entry point is plain C: pointers, sizes, PODs.  No torch / C++ types cross this line.
 * A Julia `ccall`, a Python `ctypes` or a C caller bind exactly these symbols.
 *
 * Conventions
 *  - Coordinates are (t, x, y, z); D = 4 (src/RayTraceGR.jl:253-254).
 *  - A ray state is 8 scalars (x^a, u^a) — `r2s(Ray{T}(x,u))` (src/RayTraceGR.jl:345-347).
 *  - Pixels are indexed as the reference's column-major `pixels[i,j]`: linear index i + j*ni (0-based),
 *    i fastest.  A slab is the j-range [j0, j1); its local linear index is i + (j-j0)*ni.
 *  - RGB output is three planes (SoA) of n = ni*(j1-j0) scalars each: rgb[c*n + idx]
 *    (what `colorview(RGB, R', G', B')` consumes, src/RayTraceGR.jl:566-569).
 *  - All functions return 0 on success and a negative rtgr_status on failure; they never throw or abort.
 *    The message of the last failure on the calling thread is rtgr_last_error().
 *  - The library never keeps a caller pointer after a call returns.
 *
 * Contexts, ownership, threading
 *  - All device state — workspaces, work-queue heads, run-time loaded metric modules, staging buffers, streams,
 *    timing events, launch-policy options — lives in an opaque rtgr_context created for a list of devices
 *    (rtgr_create).  Every entry point takes the context first; NULL means the process's DEFAULT context, which is
 *    created on first use on the calling thread's current HIP device (or by rtgr_init).  There is no other global state.
 *  - The reference's caller is ONE thread and the parallelism is inside the call (src/RayTraceGR.jl:510-511).  The
 *    same holds here, and more: one host thread can drive every device of a context (rtgr_trace_sharded_f64 deals the
 *    image rows to all of them and collects the frame on device 0 — one call, no MPI / torch / second process), and
 *    several host threads may use one context concurrently.
 *  - Device entry points are asynchronous on the caller's stream.  Two calls on the SAME stream are ordered by the
 *    stream and share that stream's workspace; calls on DIFFERENT streams (or devices) use different workspaces and
 *    may run concurrently.  Results are bit-identical to serial execution either way (tests: two streams, two threads).
 *  - A workspace only grows.  The superseded allocation stays alive until rtgr_trim / rtgr_destroy, so a hipGraph
 *    captured earlier never dangles; growth DURING stream capture is refused with RTGR_ERR_BAD_ARG (hipMalloc cannot be
 *    captured): call rtgr_reserve_workspace before capturing.
 */
#ifndef RTGR_H
#define RTGR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): run-time units may carry OBJECTS (RTGR_USER_OBJECT) and a resolve kernel; rtgr_object.reserved became .type;
 * the device-side record layout changed in round 4 (event-record tails, RecRef) without a bump — a unit built against the
 * older headers would have loaded and overrun the workspace.  Units also carry a hash of the device headers they were built
 * from (rtgr_user_header_hash), checked at load. */
/* 4 (round 6): `objs::Vector{Object{T}}` has no length limit in the reference (src/RayTraceGR.jl:433-441, :483, :520-526) and has none
 * here any more: rtgr_scene.objects (a caller array of any length) beside the 16 inline slots; rtgr_ray_outputs.hit32; frames in
 * flight (rtgr_trace_frames_*); rtgr_scene_check runs by itself the first time a scene with user objects is traced. */
#define RTGR_ABI_VERSION 4
#define RTGR_MAX_OBJECTS 16         /* objects held INLINE in rtgr_scene.obj; a longer list goes through rtgr_scene.objects */
#define RTGR_OBJECTS_LIMIT 1048576  /* sanity bound on rtgr_scene.nobj (2^20; beyond ~10^4 the per-step cost grows as nobj / 64)  */
#define RTGR_MAX_DEVICES 16
#define RTGR_MAX_SOURCES 16         /* object sources rtgr_user_source_join joins in one call                               */
#define RTGR_MAX_SAMPLES 32         /* type tags 0 .. 31 of a unit are asked for a sample object (rtgr_user_sample)         */

/* ---- return codes -------------------------------------------------------------------------------------- */
enum rtgr_status {
    RTGR_OK = 0,
    RTGR_ERR_BAD_ARG = -1,     /* null pointer, bad enum, empty/oversized range ...                           */
    RTGR_ERR_NO_DEVICE = -2,   /* no usable HIP device; the library has NO CPU fallback                       */
    RTGR_ERR_HIP = -3,         /* a HIP runtime call failed (message has the hipError string)                 */
    RTGR_ERR_NAN_INPUT = -4,   /* NaN in an input ray — the reference's `@assert !any(isnan,…)` (:279)        */
    RTGR_ERR_NOT_INIT = -5
};

/* ---- metric: replaces the `metric` callable argument (src/RayTraceGR.jl:262-264, :274-294) -------------- */
enum rtgr_metric {
    RTGR_MINKOWSKI = 0, /* minkowski(x)                                           src/RayTraceGR.jl:262-264   */
    RTGR_KS_REF = 1,    /* kerr_schild(x) exactly AS WRITTEN, r = sqrt(rho^2-a^2)/2 + sqrt(a^2 z^2+((rho^2-a^2)/2)^2)
                           (src/RayTraceGR.jl:284); the reference hard-wires M=1, a=0 (:275-276)               */
    RTGR_KS_TRUE = 2,   /* textbook Kerr–Schild radius r^2 = (q + sqrt(q^2 + 4 a^2 z^2))/2, q = rho^2 - a^2
                           (no reference counterpart; needed for a != 0 configs)                               */
    RTGR_USER = 3       /* the metric function rtgr_scene.user_metric names (rtgr_user_metric_load); always traced
                           with the generic dual-number RHS (RTGR_METRIC_GENERIC is implied)                   */
};

/* OR-ed into rtgr_scene.metric: evaluate the geodesic RHS the way the reference does for ANY metric callable — 4-wide
 * forward duals through the metric, symmetric g / dg, 4x4 inverse, Christoffel contraction (the "generic" path of
 * DESIGN.md §4.2) — instead of the closed Kerr–Schild contraction.  Same results to rounding; ~5x the flops.  It is the
 * path a new (non-Kerr–Schild-form) metric functor would take, and the one whose executed flops equal the algorithmic
 * count of SURVEY §8(d).  Float64 and Float32 (the reference's own test runs T = Float32 through it, test/runtests.jl:37). */
#define RTGR_METRIC_GENERIC 0x100u

/* ---- objects: replaces Vector{Object{T}} (src/RayTraceGR.jl:374-428); ORDER MATTERS (:518-530) ---------- */
enum rtgr_object_kind {
    RTGR_PLANE = 1,  /* Plane{T}(time)            p[0] = time                     src/RayTraceGR.jl:394-404  */
    RTGR_SPHERE = 2, /* Sphere{T}(pos, vel, radius)  p[0..3]=pos  p[4..7]=vel  p[8]=radius   (:409-413)
                        vel is dead data in the reference (:411, "TODO: Use metric?" :416); here it is the emitter's
                        coordinate 4-velocity for the optional redshift output (rtgr_ray_outputs.redshift)       */
    RTGR_DISK = 3,   /* thin disk (no reference counterpart): p[0]=half thickness h, p[1]=r_in, p[2]=r_out;
                        distance = max(|z|-h, r_in-rho_cyl, rho_cyl-r_out) obeying the contract at :377-383    */
    RTGR_USER_OBJECT = 4 /* a NEW subtype of the reference's open `abstract type Object{T}` (:374-389): its two methods,
                        `distance(obj, pos)::T` and `objcolor(obj, pos)::SVector{3,T}`, are given as device source and
                        compiled at run time into the scene's unit (rtgr_user_unit_compile, below); `type` tells the
                        caller's own object types apart inside that source, p[0..8] are the object's fields           */
};

typedef struct rtgr_object {
    uint32_t kind; /* rtgr_object_kind */
    uint32_t type; /* RTGR_USER_OBJECT: the tag handed to rtgr_user_distance / rtgr_user_objcolor; 0 otherwise */
    double p[9];
} rtgr_object;

typedef struct rtgr_scene {
    uint32_t metric; /* rtgr_metric (| RTGR_METRIC_GENERIC) */
    uint32_t nobj;   /* length of the object list: 0..RTGR_MAX_OBJECTS in obj[], or any number (<= RTGR_OBJECTS_LIMIT) in objects[] */
    double M;        /* mass  (reference: 1, :275) */
    double a;        /* spin  (reference: 0, :276) */
    uint64_t user_metric; /* id of the run-time compiled UNIT this scene is written for — the module that carries the kernels
                             of a user metric (RTGR_USER), of user objects (RTGR_USER_OBJECT), or of both
                             (rtgr_user_metric_load / _compile, rtgr_user_unit_compile); a scene can never run with another
                             unit's kernels.  0: built-in metric and built-in objects only. */
    rtgr_object obj[RTGR_MAX_OBJECTS];
    const rtgr_object* objects; /* NULL: the list is obj[0 .. nobj).  Otherwise the WHOLE list, nobj objects in the caller's (host)
                             memory, and obj[] is not read — `objs::Vector{Object{T}}` of any length (src/RayTraceGR.jl:433-441, :483).
                             Read during the call only, like every caller pointer.  On the device the first RTGR_MAX_OBJECTS
                             objects travel in the kernels' argument block as before; a longer list sits, whole, in a device table
                             the context keeps per distinct list (uploaded when a list is first seen — blocking, microseconds; a
                             first sight during hipGraph capture is refused: trace the scene once before capturing), its spheres
                             sorted into groups of neighbours that the kernels ask before their members (option "groups").  Cost:
                             64 objects ~1.8 x, 256-512 objects ~1.5-1.85 x the 3-object frame. */
} rtgr_scene;

/* ---- solver constants (src/RayTraceGR.jl:485, :497, :510-511, :519, :528; OrdinaryDiffEq 5.38 defaults) -- */
typedef struct rtgr_solver {
    double reltol;        /* eps(T)^(3/4)                                       :485               */
    double abstol;        /* eps(T)^(3/4)                                       :485               */
    double lambda0;       /* 0                                                  :497               */
    double lambda1;       /* 100                                                :497               */
    double hit_threshold; /* 0.01                                               :519               */
    double miss_rgb[3];   /* (1,0,0)                                            :528               */
    uint32_t max_steps;   /* step-attempt cap per ray (reference: maxiters, never binds)           */
    uint32_t interp_points; /* ContinuousCallback interp_points = 10 (DiffEqBase 6.35 default)      */
} rtgr_solver;

/* ---- camera: arguments of make_canvas (src/RayTraceGR.jl:458-462) ---------------------------------------- */
typedef struct rtgr_camera {
    double pos[4];
    double widthx[4];
    double widthy[4];
    double normal[4];
} rtgr_camera;

/* ---- per-ray status byte (the reference ignores solver retcodes, :502-505; we report them) -------------- */
enum rtgr_ray_status {
    RTGR_RAY_EVENT = 0,     /* terminated by the ContinuousCallback (terminate!, :489)      */
    RTGR_RAY_LAMBDA1 = 1,   /* reached lambda1 without an event                              */
    RTGR_RAY_MAXSTEPS = 2,  /* hit max_steps                                                 */
    RTGR_RAY_DTMIN = 3,     /* step size underflow                                           */
    RTGR_RAY_NAN = 4        /* state became non-finite                                       */
};

/* ---- counters accumulated over a call (8 x uint64) -------------------------------------------------------- */
typedef struct rtgr_counters {
    uint64_t rays;
    uint64_t accepted;        /* accepted Tsit5 steps                                         */
    uint64_t rejected;        /* rejected Tsit5 step attempts                                 */
    uint64_t rhs_evals;       /* geodesic RHS evaluations actually executed                   */
    uint64_t events;          /* rays terminated by an event                                  */
    uint64_t events_interior; /* events found only by the interior dense-output sample points */
    uint64_t not_finished;    /* rays with status >= RTGR_RAY_MAXSTEPS                        */
    uint64_t reserved;
} rtgr_counters;

/* optional per-ray outputs; any member may be NULL.  Device or host pointers according to the call. */
typedef struct rtgr_ray_outputs {
    void* state_end;      /* n x 8 scalars (AoS)  — `sols.u[i]` (src/RayTraceGR.jl:516)                    */
    void* lambda_end;     /* n scalars            — `sol.t[end]` (:503)                                     */
    uint8_t* status;      /* n x rtgr_ray_status                                                            */
    uint8_t* hit;         /* n: omin of the colouring rule (0 = miss, else 1-based object index, :518-526); scenes of
                             more than 255 objects: RTGR_ERR_BAD_ARG, ask for hit32 instead                        */
    uint32_t* n_accept;   /* n: accepted steps per ray                                                      */
    uint32_t* n_reject;   /* n: rejected attempts per ray                                                   */
    void* redshift;       /* n scalars: g = (k.u_obs)/(k.u_emit), the frequency ratio observed/emitted of the light that
                             reaches the pixel (k = the ray's tangent, . = the metric's inner product).  u_obs = the
                             camera's static observer at the pixel, future-directed: -g^-1 e_t normalised (make_canvas
                             builds its past-directed rays from +g^-1 e_t, :471-472); u_emit = the hit Sphere's `vel`
                             (:411; a future-directed coordinate 4-velocity such as the examples' (1,0,0,0)) normalised
                             at the end point, or the static observer there for planes / disks; NaN where the ray hits
                             nothing or u_emit is not timelike.
                             No reference counterpart (`vel` is stored and never used, :411, :416).  Every metric
                             (built-in, run-time compiled) and both scalar types, like `Sphere{T}` and the metric
                             argument of the reference (:409-413, :302-309); n scalars of the entry point's type. */
    uint32_t* hit32;      /* n: omin as 32 bits — for object lists of any length (may be asked for beside `hit`)              */
} rtgr_ray_outputs;

/* ---- lifecycle --------------------------------------------------------------------------------------------- */
typedef struct rtgr_context rtgr_context;

/* A context over n_devices HIP devices (device_ids[k] = HIP ordinal; the same ordinal may appear more than once: each
 * entry gets streams and workspaces of its own).  device_ids == NULL: the calling thread's current device. */
int rtgr_create(const int* device_ids, int n_devices, rtgr_context** ctx_out);
int rtgr_destroy(rtgr_context* ctx);
/* number of devices of the context (negative rtgr_status on error) */
int rtgr_context_devices(rtgr_context* ctx);
/* Synchronise the context's devices and free retired workspaces (and staging buffers). */
int rtgr_trim(rtgr_context* ctx);
/* (Re)create the process's default context on HIP device `device` (< 0: the current device).  Optional: the default
 * context is otherwise created on first use. */
int rtgr_init(int device);
/* Destroy the default context. */
int rtgr_shutdown(void);
const char* rtgr_last_error(void);
/* (The library also exports `rtgr_testhook_*` symbols — host-only probes of internal layouts for the CPU tests, e.g. how a long object
 * list's spheres are grouped — and, in debug builds, `rtgr_debug_*`: neither is part of this ABI; nothing outside tests/ may bind them.) */
int rtgr_abi_version(void);
/* Fill `s` with the reference's constants for T = Float64 (is_f32 = 0) or Float32 (is_f32 = 1). */
int rtgr_solver_defaults(rtgr_solver* s, int is_f32);
/* Name / CU count / clock of device `index` of the context, for bench reports. */
int rtgr_device_info(rtgr_context* ctx, int index, char* name, uint64_t name_len, int* n_cu, int* clock_mhz, int* wavefront);

/* Launch-policy options (experiments and schedule-invariance tests).  Names: "waves_per_cu", "waves_per_cu_near", "chunk",
 * "split", "order", "fair", "near_early", "far4", "rounds", "handback_after", "qchunk", "qchunk_near", "host_chunk", "peer" (multi-device gather:
 * 0 = through the host, 1 = peer copies or fail) — these decide WHEN and WHERE a ray is integrated and never change a result
 * bit —, and three that select another FORMULATION of the same algorithm, with results equal up to rounding: "tile" (1: the
 * simple tile-per-wave kernel), "pack" (Float32: 0 = one ray per lane, 1 = two rays per lane in packed arithmetic) and "packfar"
 * (Float32 experiment, default 0: 1 = the packed kernel without its scan as a FAR pass + the scalar NEAR pass; measured slower).
 * value -1 = automatic.  Initial values come from the environment variables RTGR_<NAME> read ONCE when the context is created.
 * "scene_check" (default 1): the automatic first-trace check of scenes with user objects (see "user objects" below); 0 = off.
 * "groups" (default 1): object lists longer than RTGR_MAX_OBJECTS have their spheres sorted into groups of neighbours, each with a
 * bounding sphere that the per-step reach test asks first, and the event root-find narrows the list to the objects the event's step
 * can meet; 0 = every object is asked every time (A/B and tests: the frames are equal bit for bit); a value >= 2 = that many spheres
 * per group at most instead of 8 (experiments: 8 is the measured optimum for 32-256 objects).  In Float64 no option changes
 * a result bit except "tile" / "pack"; in Float32 "split" (and "groups", "rounds" through it) selects kernels whose compiled
 * arithmetic differs in the last bit, as "pack" does — Float32 lists of 32 objects and more run the FAR + NEAR pair by default.
 * Two options govern how run-time units are LOADED: "unit_audit" (default 1; 0 = skip the audit of the code object for the
 * compiler's EXEC-flip fault) and "unit_probe" (default 1; 0 = skip the load-time probe) — test hooks, see rtgr_user_metric_load. */
int rtgr_set_option(rtgr_context* ctx, const char* name, long value);
int rtgr_get_option(rtgr_context* ctx, const char* name, long* value);

/* The device entry points never allocate once the stream's workspace (start / hand-over / event records, per-ray meta,
 * queue order: 213 B per ray, 373 B when end states are asked for; bounded by a pipeline chunk of 2^26 rays — fewer when that would not fit a quarter of
 * the free device memory) is large enough.  Call this once up front (e.g. before hipGraph capture) to size the workspace
 * of `stream` on the device that owns `d_any` (any device pointer of the later call; NULL: device 0 of the context). */
int rtgr_reserve_workspace(rtgr_context* ctx, const void* d_any, void* stream, uint64_t n_rays, int with_state_end, int is_f32);

/* Optional per-kernel timing for benchmarks: when enabled the library brackets every kernel it launches on device
 * `index` with HIP events on the launch stream.  rtgr_timing_read waits for them and returns, since the previous read,
 * the summed milliseconds and launch counts of [0] the ray set-up and queue-order kernels, [1] the integrate kernel's main
 * pass (FAR, or FULL when the far/near split is off — the hot kernel), [2] the resolve kernel, [3] the integrate kernel's
 * NEAR pass.  Not for use during hipGraph capture. */
int rtgr_timing_enable(rtgr_context* ctx, int index, int on);
int rtgr_timing_read(rtgr_context* ctx, int index, double ms[4], uint64_t launches[4]);
/* … and of the multi-device exchange of rtgr_trace_sharded_device_* on device `index`, since the previous read: [0] this device's
 * rows leaving it for device 0 (the peer copy, or the device -> pinned host leg of the fallback), measured on the source device's
 * stream behind its trace; [1] device 0 only: the placement kernels that put the ranks' rows back into the frame. */
int rtgr_timing_read_exchange(rtgr_context* ctx, int index, double ms[2], uint64_t launches[2]);
/* Peer access between device 0 and device `index` of the context as rtgr_create established it: 1 = peer copies (or the same
 * physical device, or index 0), 0 = not available — the gather then stages that device's rows through pinned host memory — with
 * the reason copied into `why` (may be NULL).  Negative rtgr_status on error. */
int rtgr_peer_access(rtgr_context* ctx, int index, char* why, uint64_t why_len);

/* ---- the hot path, device-resident buffers ------------------------------------------------------------------
 * Replaces the body of trace_rays (src/RayTraceGR.jl:482-536) for rows j in [j0, j1) of an ni x nj canvas.
 *   d_state0 : n x 8 initial ray states on the DEVICE (n = ni*(j1-j0)), as `input_func(i)` yields (:492-496),
 *              or NULL — then rays are generated on the device from `cam` exactly as make_canvas does (:464-476).
 *   d_rgb    : 3*n scalars on the device, plane-major (required).  The device of the context that owns this
 *              pointer runs the call.
 *   out      : optional per-ray device outputs.
 *   d_counters : optional device pointer to one rtgr_counters; the call ADDS to it (caller zeroes it).
 *   stream   : hipStream_t to enqueue on (NULL = default stream).  The call is asynchronous: it only enqueues.
 */
int rtgr_trace_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* d_state0,
                          const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1,
                          double* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);
int rtgr_trace_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* d_state0,
                          const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1,
                          float* d_rgb, const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);

/* Strided rows: traces image rows j0, j0+jstride, …, j0+(nrows-1)*jstride of the camera's ni x nj canvas (rays generated
 * on the device); local row k of the output planes (n = ni*nrows) is image row j0 + k*jstride.  jstride = 1 is a
 * contiguous slab; jstride = N, j0 = rank is the CYCLIC row split used for multi-GPU runs — contiguous slabs of a
 * black-hole image are unbalanced (rows through the hole cost ~1.8x the edge rows), cyclic rows are not. */
int rtgr_trace_rows_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                               uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, double* d_rgb,
                               const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);
int rtgr_trace_rows_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                               uint64_t ni, uint64_t nj, uint64_t j0, uint64_t jstride, uint64_t nrows, float* d_rgb,
                               const rtgr_ray_outputs* out, rtgr_counters* d_counters, void* stream);

/* ---- the hot path, host buffers (what a Julia ccall passes) ---------------------------------------------------
 * Same semantics with HOST pointers; the library stages through pinned buffers of the context, pipelines H2D copy /
 * integration / D2H copy of successive pieces on three streams, and blocks until done.
 *   state0 may be NULL (device-side make_canvas from `cam`).  `ctr` (host, optional) is overwritten.
 * EVERY device of the context takes part (the reference's caller is `trace_rays(metric, objs, canvas)`,
 * src/RayTraceGR.jl:483-484, :560, :596 — one call, parallel inside): the rows of the slab are dealt cyclically, device k
 * of N takes slab rows k, k+N, …; each device is driven by a host thread of its own inside the call, uploads ITS rows from
 * the caller's array and downloads them straight back into it over its own PCIe link (no hop through device 0, no peer
 * copies); counters are summed.  The results do not depend on the number of devices, bit for bit.  A context of one device
 * (the default context) runs on the calling thread.
 */
int rtgr_trace_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* state0,
                   const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, double* rgb,
                   const rtgr_ray_outputs* out, rtgr_counters* ctr);
int rtgr_trace_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* state0,
                   const rtgr_camera* cam, uint64_t ni, uint64_t nj, uint64_t j0, uint64_t j1, float* rgb,
                   const rtgr_ray_outputs* out, rtgr_counters* ctr);

/* Accepts the reference's own pixel array: `pointer(c.pixels)` of an Array{Pixel{Float64},2} — 11 doubles per
 * pixel (pos 4, normal 4, rgb 3; src/RayTraceGR.jl:446-450), column-major ni x nj.  Traces every pixel and
 * writes rgb back into the same AoS layout of `pixels_out` (may alias pixels_in), as trace_rays does (:532). */
int rtgr_trace_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* pixels_in,
                          uint64_t ni, uint64_t nj, double* pixels_out, rtgr_counters* ctr);
/* ... of an Array{Pixel{Float32},2}: 11 floats (44 bytes) per pixel — `Canvas{T}` is generic in T (:452-455). */
int rtgr_trace_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* pixels_in,
                          uint64_t ni, uint64_t nj, float* pixels_out, rtgr_counters* ctr);

/* ---- several frames in one call, two in flight ------------------------------------------------------------------------
 * AN EXTENSION: the reference renders one frame per call (example1 / example2, src/RayTraceGR.jl:560, :596).  A render loop delivers
 * frame after frame, and every frame's pipeline ends thin (the last rays of the FAR pass, the long stayers of the NEAR pass, the
 * last download); with two frames in flight the thin end of one overlaps the start of the next — 8-11 % per frame at 1024², 5 % for
 * one GPU's share of a 4096² frame split eight ways.  A caller that owns two HIP streams gets that from the
 * device entry points; these calls give it to the blocking ones: nframes frames of ONE scene and canvas size, frame k from cams[k]
 * (rays generated on the device), or from state0s[k] (ni*nj x 8 ray states; state0s may be NULL, and so may single entries when cams
 * is given), into rgb[k] (3 planes of ni*nj), outs[k] (outs may be NULL) and ctrs[k] (may be NULL).  The library alternates the
 * frames between two pipelines of its own — staging buffers, streams and workspace each — on every device of the context (the rows of
 * EVERY frame are dealt to all devices, as in rtgr_trace_f64).  Frame k's results are those of the single call, bit for bit.
 * Blocking; returns the first failure with the frame's number in rtgr_last_error().  The _pixels twins take and fill the reference's
 * own Array{Pixel{T},2} per frame, as rtgr_trace_pixels_f64 does. */
int rtgr_trace_frames_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams,
                          const double* const* state0s, uint64_t ni, uint64_t nj, double* const* rgb, const rtgr_ray_outputs* outs,
                          rtgr_counters* ctrs);
int rtgr_trace_frames_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes, const rtgr_camera* cams,
                          const float* const* state0s, uint64_t ni, uint64_t nj, float* const* rgb, const rtgr_ray_outputs* outs,
                          rtgr_counters* ctrs);
int rtgr_trace_frames_pixels_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes,
                                 const double* const* pixels_in, uint64_t ni, uint64_t nj, double* const* pixels_out, rtgr_counters* ctrs);
int rtgr_trace_frames_pixels_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, uint32_t nframes,
                                 const float* const* pixels_in, uint64_t ni, uint64_t nj, float* const* pixels_out, rtgr_counters* ctrs);

/* Legacy single-ray shape `trace_ray(metric, objs, cb, p)::Pixel` (test/runtests.jl:76): one pixel in, rgb out. */
int rtgr_trace_one_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double pos[4],
                       const double normal[4], double rgb[3], double state_end[8], uint8_t* status);
int rtgr_trace_one_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float pos[4],
                       const float normal[4], float rgb[3], float state_end[8], uint8_t* status);

/* ---- the hot path over ALL devices of the context, frame assembled in DEVICE memory ---------------------
 * rtgr_trace_sharded_device_*: one blocking call from one host thread: image rows are dealt cyclically to the context's N
 * devices (device k traces rows k, k+N, …), every device runs the pipeline on a stream of its own, and the rows — RGB
 * planes and, when asked for, status / hit / step counts / end states / lambda_end / redshift — are copied peer-to-peer
 * (hipMemcpyPeerAsync on the source device's stream: one xGMI link per peer, no reduction, no halo) to device 0 and put
 * back in place there: d_rgb (3 planes of ni*nj) and the members of `out` are device-0 pointers.  Counters of all devices
 * are summed into `ctr` (host).
 *   Peer access is established by rtgr_create.  Where it could not be (the reason is kept), the rows of that device travel
 * device -> pinned host -> device 0 instead; option "peer" = 0 forces that path for every device (also between two entries
 * of one physical GPU: how it is tested on a one-GPU box), "peer" = 1 makes the call fail with RTGR_ERR_HIP naming the
 * device pair instead of falling back.  A failing hipMemcpyPeerAsync is RTGR_ERR_HIP with the pair in the message.
 * rtgr_trace_sharded_f64 / _f32 (host destination) need no gather: they are rtgr_trace_f64 / _f32 with state0 = NULL over
 * the whole canvas (every device downloads its own rows). */
int rtgr_trace_sharded_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                           uint64_t ni, uint64_t nj, double* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int rtgr_trace_sharded_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt,
                                  const rtgr_camera* cam, uint64_t ni, uint64_t nj, double* d_rgb,
                                  const rtgr_ray_outputs* out, rtgr_counters* ctr);
int rtgr_trace_sharded_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam,
                           uint64_t ni, uint64_t nj, float* rgb, const rtgr_ray_outputs* out, rtgr_counters* ctr);
int rtgr_trace_sharded_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt,
                                  const rtgr_camera* cam, uint64_t ni, uint64_t nj, float* d_rgb,
                                  const rtgr_ray_outputs* out, rtgr_counters* ctr);

/* ---- camera: make_canvas (src/RayTraceGR.jl:457-478) on the device ------------------------------------------
 * Writes n x 8 ray states (pos, null past-directed 4-velocity) for rows [j0, j1).  Device / host variants. */
int rtgr_make_canvas_device_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni,
                                uint64_t nj, uint64_t j0, uint64_t j1, double* d_state0, void* stream);
int rtgr_make_canvas_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj,
                         uint64_t j0, uint64_t j1, double* state0);
/* T = Float32 (make_canvas is generic in T, :457-462) */
int rtgr_make_canvas_device_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni,
                                uint64_t nj, uint64_t j0, uint64_t j1, float* d_state0, void* stream);
int rtgr_make_canvas_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_camera* cam, uint64_t ni, uint64_t nj,
                         uint64_t j0, uint64_t j1, float* state0);

/* ---- physics kernels exposed for parity tests (test/runtests.jl:12-61 exercises exactly these) -------------
 * Evaluated ON THE DEVICE for n points (host pointers in/out):
 *   g   : n x 16  metric g_ab            (minkowski / kerr_schild, :262-294)
 *   dg  : n x 64  dg[a][b][c] = d_c g_ab (dmetric, :302-313)
 *   Gam : n x 64  Gamma^a_bc             (christoffel, :321-331)
 * Any output may be NULL.  _f32: T = Float32, as the reference's Kerr-Schild testset runs it (test/runtests.jl:37). */
int rtgr_eval_metric_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* x /* n x 4 */, uint64_t n, double* g,
                         double* dg, double* Gam);
int rtgr_eval_metric_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* x /* n x 4 */, uint64_t n, float* g,
                         float* dg, float* Gam);
/* geodesic RHS (src/RayTraceGR.jl:358-370): n x 8 states -> n x 8 derivatives, on the device.
 * path = 0: Kerr–Schild-form closed contraction with IEEE division (the tile kernel's RHS);
 * path = 1: generic dual-number path (RTGR_METRIC_GENERIC, user metrics);
 * path = 2: EXACTLY the function the production integrate loop calls (closed contraction with the fast reciprocal /
 *           reciprocal-square-root sequences, null-congruence shortcuts of the textbook metric). */
int rtgr_eval_geodesic_f64(rtgr_context* ctx, const rtgr_scene* scene, const double* s /* n x 8 */, uint64_t n, int path,
                           double* ds /* n x 8 */);
int rtgr_eval_geodesic_f32(rtgr_context* ctx, const rtgr_scene* scene, const float* s /* n x 8 */, uint64_t n, int path,
                           float* ds /* n x 8 */);
/* objects and the colour rule at n points, on the device (host pointers): `distance(obj, x)` of every object of the scene in scene order
 * (src/RayTraceGR.jl:377-419; user objects: the unit's rtgr_user_distance), `min_distance(objs, s)` (:433-441) and the colouring loop of
 * trace_rays (:513-533: nearest object below opt->hit_threshold, objcolor x omin / length(objs), else opt->miss_rgb) evaluated at x as
 * if a ray had ended there.  x: n x 4;  d: n x nobj;  dmin: n;  hit: n (omin, 0 = miss);  rgb: n x 3 (AoS).  Any output may be NULL. */
int rtgr_eval_objects_f64(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const double* x /* n x 4 */, uint64_t n,
                          double* d, double* dmin, uint8_t* hit, double* rgb);
int rtgr_eval_objects_f32(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const float* x /* n x 4 */, uint64_t n,
                          float* d, float* dmin, uint8_t* hit, float* rgb);
/* the hot loop's reciprocal and reciprocal-square-root sequences (hardware seed + one third-order correction) on n
 * operands; either output may be NULL */
int rtgr_eval_fastmath_f64(rtgr_context* ctx, const double* x, uint64_t n, double* rcp, double* rsq);

/* ---- user metrics: "metric is ANY callable x -> g" (src/RayTraceGR.jl:302-309, :358-370, :457-511) ------------
 * The reference accepts any Julia function as the metric and differentiates it with forward duals.  The native
 * counterpart: the caller writes the metric once as a C++ function template over the scalar type
 *     template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]);
 * (S = double / float for make_canvas' normalisation, S = 4-wide forward dual of either for dmetric), pastes it into
 * raytracegr.jl_amd/csrc/rtgr_user_unit.hip.in, compiles that unit with
 *     hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 -O3 -std=c++17 -I<csrc> unit.hip -o metric.hsaco
 * and hands the code object to the library, which loads it on every device of the context and returns its id (a hash
 * of the code object: loading the same file twice yields the same id and one resident copy).  A scene selects it with
 * metric = RTGR_USER and user_metric = id, in every entry point that takes a scene (trace, make_canvas, eval_metric,
 * eval_geodesic path 1); M and a of the scene are passed through to the function.  Several metrics may be resident at
 * once.  The Python mirror automates the steps (api.UserMetric).
 *
 * A metric OF KERR–SCHILD FORM, g = eta + f k (x) k with k_t = 1, k null with respect to eta and no t-dependence, may be given
 * by its two ingredients instead of its 16 entries:
 *     template <class S> __device__ void rtgr_user_ks(const S x[4], double M, double a, S& f, S k[3]);
 * (unit built with -DRTGR_USER_KS=1 -DRTGR_USER_NE=3; rtgr_user_metric_compile detects the name in the source text).  The
 * unit derives the 16 entries for make_canvas and the evaluation hooks; its integrate kernels differentiate the four scalars
 * (f, k_x, k_y, k_z) and use the closed contraction of the built-in metrics — no 4x4 solve.
 * rtgr_eval_geodesic_f64(path = 2) on a user scene evaluates exactly that loop RHS. */
int rtgr_user_metric_load(rtgr_context* ctx, const char* code_object_path, uint64_t* id_out);
/* The same in ONE call from source text: `source` (the definition of rtgr_user_metric<S>, as above) is pasted into the unit
 * template, compiled IN-PROCESS and loaded — no hipcc on the box, no child process, ~10 s for a light metric.  The build goes
 * hiprtc (source -> optimised bitcode) -> libamd_comgr (bitcode -> assembly listing -> code object), both resolved with dlopen at
 * first use, with a look at the LISTING in between: the check and repair described below.  stationary != 0 declares that the metric
 * does not depend on t (-DRTGR_USER_NE=3: the integrate kernels carry the three spatial partials only).  A unit whose integrate
 * kernels would spill is rebuilt with more registers per lane.  The unit template and device headers are read from the `csrc`
 * directory next to librtgr_hip.so (or $RTGR_CSRC).  On a compile error the compiler's log is the rtgr_last_error(). */
int rtgr_user_metric_compile(rtgr_context* ctx, const char* source, int stationary, uint64_t* id_out);
/* The build step of rtgr_user_metric_compile on its own: source text -> code object FILE for rtgr_user_metric_load (build once,
 * keep the file, load it in every later process).  Needs no GPU and no context. */
int rtgr_user_metric_build(const char* source, int stationary, const char* code_object_path);
int rtgr_user_metric_unload(rtgr_context* ctx, uint64_t id); /* id 0: all */

/* ---- user objects: `Object{T}` is an OPEN abstract type (src/RayTraceGR.jl:374-389) ------------------------------------------
 * The reference's second extension point: any `struct MyThing{T} <: Object{T}` with the two methods
 *     distance(obj, pos::SVector{D,T})::T                "zero on the surface, positive outside, negative inside"   (:377-386)
 *     objcolor(obj, pos::SVector{D,T})::SVector{3,T}                                                                 (:387-389)
 * is traced — the ContinuousCallback condition min_distance dispatches on it (:433-441) and so does the colour rule (:518-530).
 * The native counterpart mirrors user metrics: the two methods are written once as device function templates over the scalar
 *     template <class S> __device__ S    rtgr_user_distance(unsigned type, const S x[4], const S p[9]);
 *     template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]);
 * (S = double or float; `type` and `p` are the rtgr_object's: one source may define several object types and switch on `type`;
 * the m* helpers of user metrics — msqrt mabs macos matan2 ... — are available), and compiled at run time into the scene's UNIT,
 * the code object that carries every kernel that calls them: ray set-up (the condition's initial sign), the FAR / NEAR / FULL
 * integrate passes (the sample-point scan) and the resolve kernel (event root-find, colour rule).  Objects are independent of
 * the metric in the reference; in compiled code the integrate kernels contain both, so a unit is built for ONE metric variant:
 * its own (the same source also defines rtgr_user_metric / rtgr_user_ks), or a built-in one, named by `built_for` — the scene
 * the unit is meant for: its metric enum, RTGR_METRIC_GENERIC flag and whether a != 0 select the kernels' instantiation; its
 * objects and the values of M and a do not matter — exactly what Julia specialises `solve` on.  A scene whose metric variant
 * differs from the unit's is refused (RTGR_ERR_BAD_ARG), never traced with the wrong kernels.
 *   Optional third function: the FAR pass skips the scan of a step when no object's distance can change sign within the
 * step's reach; for a user object it needs a bound from the source,
 *     template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]);
 * = an upper bound of |distance(x') - distance(x)| over all x' with |x'_q - x_q| <= dl[q].  Without it ("rtgr_user_reach" does
 * not occur in the source) a user object is never provably out of reach and scenes of the unit run the single FULL pass
 * (every accepted step scanned, as the reference does): same results, the FAR pass's saving is lost.
 *   Optional fourth function — a SAMPLE OBJECT per type for the library's own checks:
 *     template <class S> __device__ bool rtgr_user_sample(unsigned type, S p[9]);
 * = true with p filled for an object of `type` about 0.5 across placed around (x, y, z) = (4, 0, 0) — where example2's small sphere
 * stands, in view of the probe's camera —, false for a type the source has none for.  With it the load-time probe (below) traces
 * the unit's OWN objects through its FULL and FAR + NEAR passes, so a distance / reach pair that disagrees on the samples keeps the
 * unit from loading at all; without it the probe's frame holds built-in objects only and a bad bound is left to the scene check.
 *   Whether or not a source brings samples: the FIRST trace of every scene that holds user objects with a reach bound runs
 * rtgr_scene_check's comparison by itself, on a coarse sample of that call's own rays (option "scene_check", default 1; a few
 * milliseconds once per (unit, object list, metric parameters, solver constants, camera), answered from a table afterwards; not
 * during hipGraph capture) — a reach function that is not an upper bound is refused with RTGR_ERR_BAD_ARG instead of losing hits.
 *   Scenes: obj[k].kind = RTGR_USER_OBJECT, .type, .p as the source expects; scene.user_metric = the unit's id.  Built-in
 * objects may stand beside user objects in any order (:518-530's order rule applies to all).  Limits: the tile kernel
 * (option tile = 1) and the packed Float32 kernel know no user objects — scenes of a unit take the pipeline's scalar kernels;
 * the redshift output treats a user object's emitter as the static observer (as for planes and disks).
 * rtgr_user_metric_compile(ctx, source, stationary, id) == rtgr_user_unit_compile(ctx, source, stationary, NULL, id).
 *   Environment: RTGR_UNIT_CACHE=<directory> (opt-in) keeps the code objects rtgr_user_unit_compile / rtgr_user_metric_compile build,
 * keyed by source text + what they are built for + the device headers; a later process with the same inputs loads the file (audited
 * and probed like any other) instead of compiling for seconds.  Within a process the same call again — same context, source,
 * `stationary` and metric variant — is answered at once with the id it gave before, for as long as that unit is resident. */
int rtgr_user_unit_compile(rtgr_context* ctx, const char* source, int stationary, const rtgr_scene* built_for, uint64_t* id_out);
/* ... the build step on its own (no GPU, no context): source -> code object file for rtgr_user_metric_load */
int rtgr_user_unit_build(const char* source, int stationary, const rtgr_scene* built_for, const char* code_object_path);
/* Objects of SEVERAL sources in one scene (the reference puts any mix of Object subtypes into `objs`; compiled code holds a scene's
 * objects in one unit): joins n object sources into ONE source text for rtgr_user_unit_compile / _build.  Source k, which defines
 * ntypes[k] object types (tags 0 .. ntypes[k]-1), is wrapped in a namespace of its own; the joined text's methods dispatch on the tag:
 * type t of source k is type ntypes[0] + ... + ntypes[k-1] + t of the joined text — the caller adds that base to obj[].type.  Where
 * some sources bring rtgr_user_reach and others do not, the joined bound is +infinity for the types of the latter: every
 * step of a scene that holds such an object is scanned (the NEAR pass), scenes without one keep the FAR pass's saving.  A metric's
 * source is not joined: put it before the joined text, as with one family.  Text in, text out — no GPU, no context.
 *   `out` may be NULL (then only *need, the length with its NUL, is written); a buffer shorter than *need is RTGR_ERR_BAD_ARG. */
int rtgr_user_source_join(const char* const* sources, const uint32_t* ntypes, int n, char* out, uint64_t cap, uint64_t* need);
/* What a resident unit was built for. */
typedef struct rtgr_unit_info {
    uint32_t metric;       /* rtgr_metric (| RTGR_METRIC_GENERIC) its kernels are instantiated for; RTGR_USER: its own */
    uint32_t spin;         /* built-in closed-form kernels: 1 = the a != 0 instantiation                               */
    uint32_t has_objects;  /* the source defined rtgr_user_distance / rtgr_user_objcolor                               */
    uint32_t has_reach;    /* ... and rtgr_user_reach: scenes with its objects run FAR + NEAR                          */
    uint32_t far_waves, near_waves, f32_waves; /* waves per SIMD its integrate passes were built for                  */
    uint32_t probe_ok;     /* 1: the load-time probe ran (FULL == FAR + NEAR, two schedules each); 0: skipped (unit_probe = 0) */
} rtgr_unit_info;
int rtgr_user_unit_info(rtgr_context* ctx, uint64_t id, rtgr_unit_info* info);
/* The library's central schedule-invariance property, as a check a caller can run on ITS scene: the FAR + NEAR passes skip the
 * ContinuousCallback scan of a step only where no object's distance can change sign, so they must deliver the frame of the single
 * FULL pass (every accepted step scanned, as the reference does, src/RayTraceGR.jl:488-490).  For built-in objects the bound is the
 * library's; for user objects it is the source's rtgr_user_reach — and a reach function that is NOT an upper bound loses hits
 * silently.  This traces the camera's canvas at ni x nj rays (<= 256 x 256; a coarse canvas of the same camera is the usual call)
 * through both pass structures on device 0 of the context and compares: bit for bit for a built-in metric (with or without user
 * objects), within the load-time probe's bars for a metric given as source.  RTGR_OK, or RTGR_ERR_BAD_ARG with the count of rays
 * that differ in rtgr_last_error().  Blocking; a few milliseconds.  is_f32 must be 0 (Float32 runs one pass structure only). */
int rtgr_scene_check(rtgr_context* ctx, const rtgr_scene* scene, const rtgr_solver* opt, const rtgr_camera* cam, uint64_t ni, uint64_t nj, int is_f32);
/* The EXEC-flip check.  ROCm 7.2's compiler can place a register copy or spill at the top of the FLOW block of a divergent if / else,
 * AHEAD of the instruction that switches EXEC to the `else` lanes; code of that shape computes wrong values in some lanes (DESIGN.md
 * §4.6: the Float64 FULL pass of a heavy metric was wrong from it in round 4).  rtgr_user_metric_compile / _build look for the shape
 * in the listing and rewrite it there (as raytracegr.jl_amd/user_metric.py does on the hipcc route); rtgr_user_metric_load AUDITS
 * every code object it is handed — whatever built it — and refuses one that carries the shape with RTGR_ERR_BAD_ARG and the offending
 * instructions in rtgr_last_error().  rtgr_code_object_audit is that audit on its own (no GPU, no context needed): *found = number
 * of such blocks, their description in `report` (may be NULL).  The file is a gfx950 code object, or a host library that embeds
 * code objects — librtgr_hip.so itself: the build checks audit the kernels the library ships.  RTGR_ERR_BAD_ARG when it is neither,
 * or the disassembler (libamd_comgr) is not on the box. */
int rtgr_code_object_audit(const char* code_object_path, int* found, char* report, uint64_t report_len);
/* … and the check / repair of a gfx950 assembly LISTING (`hipcc -S`), the step rtgr_user_metric_build runs between code generation and
 * the assembler: repaired_path == NULL: *blocks = FLOW blocks that carry the shape; otherwise the listing is rewritten into
 * repaired_path and *blocks = blocks rewritten (RTGR_ERR_BAD_ARG, nothing written, when a block is not of the form the rewrite is
 * proven for).  The same rule as raytracegr.jl_amd/isa_exec.py; the tests hold the two to the same answers. */
int rtgr_listing_repair(const char* listing_path, const char* repaired_path, int* blocks);
/* Every unit that rtgr_user_metric_load / _compile / rtgr_user_unit_compile brings in is checked THREE ways before a scene can use
 * it: (1) ABI version and — when the builder recorded it — a hash of the device headers it was compiled against, which must be the
 * one the library's own kernels were built from (the record layouts the kernels share are not part of this header and move
 * without the ABI version moving); (2) the audit above, on bare code objects and on clang offload bundles alike; an image the
 * audit cannot read is refused, a box without the disassembler loads unaudited; (3) a PROBE: the unit traces a fixed 32 x 32 frame
 * through its single FULL pass and through its FAR + NEAR passes, each under TWO schedules (sixteen waves with one ray per lane;
 * three waves whose lanes refill from the queue — a ray's bits never depend on the schedule), the Float32 FULL pass likewise, and
 * is refused (RTGR_ERR_BAD_ARG, nothing left resident) when the two schedules of one structure differ in any bit or the structures
 * disagree — the symptoms of a mis-compiled unit, whatever the instruction shape (~10-130 ms). */
/* 1 if module `id` is resident (id 0: any module), else 0 */
int rtgr_user_metric_loaded(rtgr_context* ctx, uint64_t id);

/* ---- image output: N0f8 quantisation + transposed PNG layout of `save(file, colorview(...))` (:566-575) ---- */
/* rgb planes (n = ni*nj, device pointer) -> 8-bit interleaved image[j][i][c] (device pointer, 3*n bytes). */
int rtgr_quantize_device_f64(rtgr_context* ctx, const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTGR_H */
