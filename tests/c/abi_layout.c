/* abi_layout.c — a plain C caller of include/rtgr.h, compiled with gcc by tests/test_abi.py.
 *
 * What it pins: the byte layout a Julia `ccall` depends on.  julia/RayTraceGRHIP.jl passes isbits structs
 * (RtgrScene with an NTuple{16,RtgrObject}, RtgrSolver, RtgrCounters) by Ref and `pointer(c.pixels)` of an
 * Array{Pixel{Float64},2}; Julia lays isbits structs out by the C rules, so the numbers below ARE the `fieldoffset`
 * table of the Julia stub (mirrored there as a comment block).  Every offset is a _Static_assert: a header edit that
 * moves a field fails this file at compile time, on a box with no GPU.
 *
 *   abi_layout --symbols <lib>      dlopen the library and resolve every entry point the stub binds (no GPU needed)
 *   abi_layout --render  <lib> out  example2() at 200x200 the way the reference does it — make_canvas, pack the rays into an
 *                                   88-byte Pixel array exactly as src/RayTraceGR.jl:446-450 lays it out, rtgr_trace_pixels_f64
 *                                   — and write the N0f8 image[j][i][c] bytes (compared with sphere2.png by the test);
 *                                   also rtgr_trace_one_f64 on the centre pixel (legacy trace_ray shape, test/runtests.jl:76)
 *   abi_layout --render  <lib> out N the same through an explicit context that lists device 0 N times (rtgr_create): the
 *                                   drop-in entry deals the canvas rows to every device of the context — what a Julia
 *                                   `trace_rays(...; ctx = Context(0:7))` does on an 8-GPU node
 *   abi_layout --render-disk <lib> out [N]
 *                                   BASELINE config 5's scene the way julia/RayTraceGRHIP.jl's `render(KerrSchild(1.0, 0.998),
 *                                   [caelum, frustum, Disk(0.05, 2, 4)], cam...; details = true)` passes it: RTGR_KS_TRUE with
 *                                   (M, a) in the scene, an RTGR_DISK object, the camera struct (rays generated on the device),
 *                                   rtgr_trace_f64 with an rtgr_ray_outputs block — 64 x 64; writes the three f64 RGB planes, then
 *                                   hit, status, n_accept, n_reject (compared with the oracle by the test)
 *   abi_layout --render-user-objects <lib> out
 *                                   the reference's SECOND extension point from plain C, no Python and no hipcc in the process: two
 *                                   separately written object sources (a torus; the reference's Sphere as device source) joined by
 *                                   rtgr_user_source_join, compiled IN-PROCESS by rtgr_user_unit_compile for example2's metric, the
 *                                   unit's id put into the scene, rtgr_user_unit_info, rtgr_scene_check, rtgr_trace_f64 at 64 x 64
 *                                   with an rtgr_ray_outputs block; same output format as --render-disk (compared with the oracle)
 *   abi_layout --render-many <lib> out N
 *                                   `objs::Vector{Object{T}}` of any length (src/RayTraceGR.jl:433-441) from plain C: N objects (sky, far
 *                                   plane, N - 2 small spheres on a spiral around the hole) in a caller array behind rtgr_scene.objects
 *                                   — what julia/RayTraceGRHIP.jl passes as `pointer(packed)` —, rtgr_trace_f64 at 48 x 48 with the
 *                                   32-bit hit map (rtgr_ray_outputs.hit32); writes the RGB planes, hit32, status, n_accept, n_reject
 *                                   and the N objects themselves (the test builds the oracle's scene from those bytes)
 */
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rtgr.h"

/* ---- the table (bytes) -------------------------------------------------------------------------------------------- */
_Static_assert(sizeof(rtgr_object) == 80, "rtgr_object");
_Static_assert(offsetof(rtgr_object, kind) == 0 && offsetof(rtgr_object, type) == 4 && offsetof(rtgr_object, p) == 8, "rtgr_object fields");
_Static_assert(sizeof(rtgr_scene) == 1320, "rtgr_scene");
_Static_assert(offsetof(rtgr_scene, metric) == 0 && offsetof(rtgr_scene, nobj) == 4 && offsetof(rtgr_scene, M) == 8 &&
               offsetof(rtgr_scene, a) == 16 && offsetof(rtgr_scene, user_metric) == 24 && offsetof(rtgr_scene, obj) == 32 &&
               offsetof(rtgr_scene, objects) == 1312, "rtgr_scene fields");
_Static_assert(sizeof(rtgr_solver) == 72, "rtgr_solver");
_Static_assert(offsetof(rtgr_solver, reltol) == 0 && offsetof(rtgr_solver, abstol) == 8 && offsetof(rtgr_solver, lambda0) == 16 &&
               offsetof(rtgr_solver, lambda1) == 24 && offsetof(rtgr_solver, hit_threshold) == 32 &&
               offsetof(rtgr_solver, miss_rgb) == 40 && offsetof(rtgr_solver, max_steps) == 64 &&
               offsetof(rtgr_solver, interp_points) == 68, "rtgr_solver fields");
_Static_assert(sizeof(rtgr_camera) == 128, "rtgr_camera");
_Static_assert(offsetof(rtgr_camera, pos) == 0 && offsetof(rtgr_camera, widthx) == 32 && offsetof(rtgr_camera, widthy) == 64 &&
               offsetof(rtgr_camera, normal) == 96, "rtgr_camera fields");
_Static_assert(sizeof(rtgr_counters) == 64, "rtgr_counters");
_Static_assert(offsetof(rtgr_counters, rays) == 0 && offsetof(rtgr_counters, accepted) == 8 && offsetof(rtgr_counters, rejected) == 16 &&
               offsetof(rtgr_counters, rhs_evals) == 24 && offsetof(rtgr_counters, events) == 32 &&
               offsetof(rtgr_counters, events_interior) == 40 && offsetof(rtgr_counters, not_finished) == 48, "rtgr_counters fields");
_Static_assert(sizeof(rtgr_ray_outputs) == 64, "rtgr_ray_outputs");
_Static_assert(offsetof(rtgr_ray_outputs, state_end) == 0 && offsetof(rtgr_ray_outputs, lambda_end) == 8 &&
               offsetof(rtgr_ray_outputs, status) == 16 && offsetof(rtgr_ray_outputs, hit) == 24 &&
               offsetof(rtgr_ray_outputs, n_accept) == 32 && offsetof(rtgr_ray_outputs, n_reject) == 40 &&
               offsetof(rtgr_ray_outputs, redshift) == 48 && offsetof(rtgr_ray_outputs, hit32) == 56, "rtgr_ray_outputs fields");

/* Pixel{Float64} of the reference (src/RayTraceGR.jl:446-450): pos::SVector{4}, normal::SVector{4}, rgb::SVector{3} */
typedef struct { double pos[4], normal[4], rgb[3]; } pixel_f64;
_Static_assert(sizeof(pixel_f64) == 88 && offsetof(pixel_f64, normal) == 32 && offsetof(pixel_f64, rgb) == 64, "Pixel{Float64}");
typedef struct { float pos[4], normal[4], rgb[3]; } pixel_f32;   /* Pixel{Float32}: what rtgr_trace_pixels_f32 takes */
_Static_assert(sizeof(pixel_f32) == 44 && offsetof(pixel_f32, normal) == 16 && offsetof(pixel_f32, rgb) == 32, "Pixel{Float32}");

static const char* const BOUND[] = {"rtgr_create", "rtgr_destroy", "rtgr_context_devices", "rtgr_init", "rtgr_shutdown", "rtgr_last_error", "rtgr_abi_version",
                                    "rtgr_solver_defaults", "rtgr_trace_pixels_f64", "rtgr_trace_pixels_f32", "rtgr_trace_one_f64", "rtgr_trace_one_f32", "rtgr_trace_f64",
                                    "rtgr_trace_sharded_f64", "rtgr_make_canvas_f64", "rtgr_user_metric_load", "rtgr_user_metric_compile", "rtgr_user_unit_compile",
                                    "rtgr_eval_metric_f64", "rtgr_eval_geodesic_f64", "rtgr_user_source_join", "rtgr_user_unit_info", "rtgr_scene_check",
                                    "rtgr_eval_objects_f64", "rtgr_eval_objects_f32", "rtgr_trace_frames_f64", "rtgr_trace_frames_f32",
                                    "rtgr_trace_frames_pixels_f64", "rtgr_trace_frames_pixels_f32", NULL};

typedef int (*fn_defaults)(rtgr_solver*, int);
typedef int (*fn_canvas)(rtgr_context*, const rtgr_scene*, const rtgr_camera*, uint64_t, uint64_t, uint64_t, uint64_t, double*);
typedef int (*fn_pixels)(rtgr_context*, const rtgr_scene*, const rtgr_solver*, const double*, uint64_t, uint64_t, double*, rtgr_counters*);
typedef int (*fn_one)(rtgr_context*, const rtgr_scene*, const rtgr_solver*, const double*, const double*, double*, double*, uint8_t*);
typedef int (*fn_trace)(rtgr_context*, const rtgr_scene*, const rtgr_solver*, const double*, const rtgr_camera*, uint64_t, uint64_t,
                        uint64_t, uint64_t, double*, const rtgr_ray_outputs*, rtgr_counters*);
typedef const char* (*fn_err)(void);
typedef int (*fn_create)(const int*, int, rtgr_context**);
typedef int (*fn_destroy)(rtgr_context*);
typedef int (*fn_ndev)(rtgr_context*);

typedef int (*fn_join)(const char* const*, const uint32_t*, int, char*, uint64_t, uint64_t*);
typedef int (*fn_unit_compile)(rtgr_context*, const char*, int, const rtgr_scene*, uint64_t*);
typedef int (*fn_unit_info)(rtgr_context*, uint64_t, rtgr_unit_info*);
typedef int (*fn_scene_check)(rtgr_context*, const rtgr_scene*, const rtgr_solver*, const rtgr_camera*, uint64_t, uint64_t, int);

/* two object families, written separately: `distance` and `objcolor` of a new Object subtype (src/RayTraceGR.jl:377-389) as device
 * source — a torus around the z axis (p = centre, R, r), and the reference's own Sphere (:409-428; p = pos, vel, radius) */
static const char* const TORUS_SRC =
    "template <class S> __device__ S rtgr_user_distance(unsigned type, const S x[4], const S p[9]) {\n"
    "    const S X = x[1] - p[0], Y = x[2] - p[1], Z = x[3] - p[2], w = msqrt(X * X + Y * Y) - p[3];\n"
    "    return w * w + Z * Z - p[4] * p[4];\n"
    "}\n"
    "template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]) {\n"
    "    const S pi = S(3.14159265358979323846264338327950288);\n"
    "    const S X = x[1] - p[0], Y = x[2] - p[1], Z = x[3] - p[2], w = msqrt(X * X + Y * Y) - p[3];\n"
    "    rgb[0] = mod1<S>(S(6) * matan2(Y, X) / pi); rgb[1] = mod1<S>(S(6) * matan2(Z, w) / pi); rgb[2] = S(0.5);\n"
    "}\n"
    "template <class S> __device__ S rtgr_user_reach(unsigned type, const S x[4], const S p[9], const S dl[4]) {\n"
    "    const S X = x[1] - p[0], Y = x[2] - p[1], Z = x[3] - p[2], w = msqrt(X * X + Y * Y) - p[3], d = msqrt(dl[1] * dl[1] + dl[2] * dl[2]);\n"
    "    return d * (S(2) * mabs(w) + d) + dl[3] * (S(2) * mabs(Z) + dl[3]);\n"
    "}\n";
static const char* const BALL_SRC =
    "template <class S> __device__ S rtgr_user_distance(unsigned type, const S x[4], const S p[9]) {\n"
    "    const S dx = x[1] - p[1], dy = x[2] - p[2], dz = x[3] - p[3], d = dx * dx + dy * dy + dz * dz - p[8] * p[8];\n"
    "    return p[8] < S(0) ? -d : d;\n"
    "}\n"
    "template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]) {\n"
    "    const S pi = S(3.14159265358979323846264338327950288);\n"
    "    const S dx = x[1] - p[1], dy = x[2] - p[2], dz = x[3] - p[3], r = msqrt(dx * dx + dy * dy + dz * dz);\n"
    "    rgb[0] = mod1<S>(S(12) * macos(dz / r) / pi); rgb[1] = mod1<S>(S(12) * matan2(dy, dx) / pi); rgb[2] = S(1);\n"
    "}\n";                                  /* (no reach bound: its objects are scanned on every step) */

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: abi_layout --symbols|--render|--render-disk|--render-user-objects|--render-many <lib> [out [ndev | nobj]]\n"); return 2; }
    void* h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    for (int i = 0; BOUND[i]; i++)
        if (!dlsym(h, BOUND[i])) { fprintf(stderr, "missing symbol %s\n", BOUND[i]); return 4; }
    if (strcmp(argv[1], "--symbols") == 0) {
        printf("scene %zu solver %zu camera %zu counters %zu outputs %zu object %zu pixel %zu\n", sizeof(rtgr_scene),
               sizeof(rtgr_solver), sizeof(rtgr_camera), sizeof(rtgr_counters), sizeof(rtgr_ray_outputs), sizeof(rtgr_object), sizeof(pixel_f64));
        return 0;
    }
    if (argc < 4) return 2;
    fn_defaults defaults = (fn_defaults)dlsym(h, "rtgr_solver_defaults");
    fn_canvas canvas = (fn_canvas)dlsym(h, "rtgr_make_canvas_f64");
    fn_pixels pixels = (fn_pixels)dlsym(h, "rtgr_trace_pixels_f64");
    fn_one one = (fn_one)dlsym(h, "rtgr_trace_one_f64");
    fn_err err = (fn_err)dlsym(h, "rtgr_last_error");
    rtgr_context* ctx = NULL;   /* NULL = the process's default context (one device) */
    const int ndev = (argc > 4 && strcmp(argv[1], "--render-many") != 0) ? atoi(argv[4]) : 0;   /* (--render-many: argv[4] is the object count) */
    if (ndev > 0) {
        int ids[RTGR_MAX_DEVICES] = {0};
        if (ndev > RTGR_MAX_DEVICES) return 2;
        if (((fn_create)dlsym(h, "rtgr_create"))(ids, ndev, &ctx)) { fprintf(stderr, "rtgr_create: %s\n", err()); return 14; }
        if (((fn_ndev)dlsym(h, "rtgr_context_devices"))(ctx) != ndev) return 15;
    }
    if (strcmp(argv[1], "--render-disk") == 0) {
        fn_trace trace = (fn_trace)dlsym(h, "rtgr_trace_f64");
        rtgr_scene sc;
        memset(&sc, 0, sizeof sc);
        sc.metric = RTGR_KS_TRUE; sc.M = 1.0; sc.a = 0.998; sc.nobj = 3;                     /* KerrSchild(1.0, 0.998) */
        sc.obj[0].kind = RTGR_SPHERE; sc.obj[0].p[4] = 1.0; sc.obj[0].p[8] = -10.0;             /* caelum */
        sc.obj[1].kind = RTGR_PLANE; sc.obj[1].p[0] = -20.0;                                     /* frustum */
        sc.obj[2].kind = RTGR_DISK; sc.obj[2].p[0] = 0.05; sc.obj[2].p[1] = 2.0; sc.obj[2].p[2] = 4.0;   /* Disk(0.05, 2, 4) */
        rtgr_camera cam;
        memset(&cam, 0, sizeof cam);
        cam.pos[1] = 4.0; cam.pos[2] = -2.0; cam.widthx[1] = 1.0; cam.widthy[3] = 1.0; cam.normal[2] = 1.0;
        rtgr_solver opt;
        if (defaults(&opt, 0)) return 5;
        const uint64_t ni = 64, nj = 64, n = ni * nj;
        double* rgb = (double*)calloc(3 * n, sizeof(double));
        uint8_t* hit = (uint8_t*)calloc(n, 1);
        uint8_t* status = (uint8_t*)calloc(n, 1);
        uint32_t* nacc = (uint32_t*)calloc(n, 4);
        uint32_t* nrej = (uint32_t*)calloc(n, 4);
        rtgr_ray_outputs outs;
        memset(&outs, 0, sizeof outs);
        outs.hit = hit; outs.status = status; outs.n_accept = nacc; outs.n_reject = nrej;
        rtgr_counters ctr;
        if (trace(ctx, &sc, &opt, NULL, &cam, ni, nj, 0, nj, rgb, &outs, &ctr)) { fprintf(stderr, "rtgr_trace_f64: %s\n", err()); return 7; }
        if (ctr.rays != n) { fprintf(stderr, "counters: %llu rays\n", (unsigned long long)ctr.rays); return 8; }
        FILE* f = fopen(argv[3], "wb");
        if (!f) return 9;
        fwrite(rgb, sizeof(double), 3 * n, f);
        fwrite(hit, 1, n, f);
        fwrite(status, 1, n, f);
        fwrite(nacc, 4, n, f);
        fwrite(nrej, 4, n, f);
        fclose(f);
        if (ctx && ((fn_destroy)dlsym(h, "rtgr_destroy"))(ctx)) return 16;
        printf("ok %llu rays %llu events\n", (unsigned long long)ctr.rays, (unsigned long long)ctr.events);
        return 0;
    }
    if (strcmp(argv[1], "--render-many") == 0) {
        if (argc < 5) return 2;
        const uint32_t nobj = (uint32_t)atoi(argv[4]);
        if (nobj < 3) return 2;
        fn_trace trace = (fn_trace)dlsym(h, "rtgr_trace_f64");
        rtgr_object* objs = (rtgr_object*)calloc(nobj, sizeof(rtgr_object));
        objs[0].kind = RTGR_SPHERE; objs[0].p[4] = 1.0; objs[0].p[8] = -12.0;                  /* caelum */
        objs[1].kind = RTGR_PLANE; objs[1].p[0] = -25.0;                                        /* frustum */
        for (uint32_t k = 2; k < nobj; k++) {                                                   /* small spheres on a spiral around the hole */
            const double t = 0.7 * (double)k, rad = 3.0 + 4.5 * (double)(k - 2) / (double)nobj;
            objs[k].kind = RTGR_SPHERE;
            objs[k].p[1] = rad * cos(t); objs[k].p[2] = rad * sin(t) + 1.5; objs[k].p[3] = 1.2 * sin(2.3 * t);
            objs[k].p[4] = 1.0; objs[k].p[8] = 0.2 + 0.25 * fabs(sin(1.1 * t));
        }
        /* the last object stands in front of the camera: hits of the highest index, whatever the length of the list */
        objs[nobj - 1].p[1] = 4.3; objs[nobj - 1].p[2] = 0.5; objs[nobj - 1].p[3] = 0.3; objs[nobj - 1].p[8] = 0.2;
        rtgr_scene sc;
        memset(&sc, 0, sizeof sc);
        sc.metric = RTGR_KS_REF; sc.M = 1.0; sc.a = 0.0;                                        /* kerr_schild as written */
        sc.nobj = nobj; sc.objects = objs;                                                      /* the whole list; obj[] is not read */
        rtgr_camera cam;
        memset(&cam, 0, sizeof cam);
        cam.pos[1] = 4.0; cam.pos[2] = -2.0; cam.widthx[1] = 1.0; cam.widthy[3] = 1.0; cam.normal[2] = 1.0;
        rtgr_solver opt;
        if (defaults(&opt, 0)) return 5;
        const uint64_t ni = 48, nj = 48, n = ni * nj;
        double* rgb = (double*)calloc(3 * n, sizeof(double));
        uint32_t* hit32 = (uint32_t*)calloc(n, 4);
        uint8_t* status = (uint8_t*)calloc(n, 1);
        uint32_t* nacc = (uint32_t*)calloc(n, 4);
        uint32_t* nrej = (uint32_t*)calloc(n, 4);
        rtgr_ray_outputs outs;
        memset(&outs, 0, sizeof outs);
        outs.hit32 = hit32; outs.status = status; outs.n_accept = nacc; outs.n_reject = nrej;
        rtgr_counters ctr;
        if (trace(ctx, &sc, &opt, NULL, &cam, ni, nj, 0, nj, rgb, &outs, &ctr)) { fprintf(stderr, "rtgr_trace_f64: %s\n", err()); return 7; }
        if (ctr.rays != n) return 8;
        sc.objects = NULL;                   /* a list beyond the inline slots without the array is refused, not truncated */
        if (nobj > RTGR_MAX_OBJECTS && trace(ctx, &sc, &opt, NULL, &cam, ni, nj, 0, nj, rgb, NULL, NULL) == 0) { fprintf(stderr, "a truncated list was traced\n"); return 26; }
        FILE* f = fopen(argv[3], "wb");
        if (!f) return 9;
        fwrite(rgb, sizeof(double), 3 * n, f);
        fwrite(hit32, 4, n, f);
        fwrite(status, 1, n, f);
        fwrite(nacc, 4, n, f);
        fwrite(nrej, 4, n, f);
        fwrite(objs, sizeof(rtgr_object), nobj, f);
        fclose(f);
        printf("ok %llu rays %u objects\n", (unsigned long long)ctr.rays, nobj);
        return 0;
    }
    if (strcmp(argv[1], "--render-user-objects") == 0) {
        fn_trace trace = (fn_trace)dlsym(h, "rtgr_trace_f64");
        fn_join join = (fn_join)dlsym(h, "rtgr_user_source_join");
        fn_unit_compile compile = (fn_unit_compile)dlsym(h, "rtgr_user_unit_compile");
        fn_unit_info info = (fn_unit_info)dlsym(h, "rtgr_user_unit_info");
        fn_scene_check check = (fn_scene_check)dlsym(h, "rtgr_scene_check");
        if (!trace || !join || !compile || !info || !check) return 4;
        /* the two sources become one: torus = type 0 of family 0 -> 0, ball = type 0 of family 1 -> ntypes[0] + 0 = 1 */
        const char* const srcs[2] = {TORUS_SRC, BALL_SRC};
        const uint32_t ntypes[2] = {1, 1};
        uint64_t need = 0;
        if (join(srcs, ntypes, 2, NULL, 0, &need) || need == 0) { fprintf(stderr, "rtgr_user_source_join: %s\n", err()); return 20; }
        char* joined = (char*)malloc(need);
        if (join(srcs, ntypes, 2, joined, need, &need)) { fprintf(stderr, "rtgr_user_source_join: %s\n", err()); return 21; }
        rtgr_scene sc;
        memset(&sc, 0, sizeof sc);
        sc.metric = RTGR_KS_REF; sc.M = 1.0; sc.a = 0.0; sc.nobj = 4;                        /* kerr_schild as written */
        sc.obj[0].kind = RTGR_SPHERE; sc.obj[0].p[4] = 1.0; sc.obj[0].p[8] = -10.0;           /* caelum */
        sc.obj[1].kind = RTGR_PLANE; sc.obj[1].p[0] = -20.0;                                   /* frustum */
        sc.obj[2].kind = RTGR_USER_OBJECT; sc.obj[2].type = 0;                                 /* Torus(centre (4, 0, 0), R 0.9, r 0.3) */
        sc.obj[2].p[0] = 4.0; sc.obj[2].p[3] = 0.9; sc.obj[2].p[4] = 0.3;
        sc.obj[3].kind = RTGR_USER_OBJECT; sc.obj[3].type = ntypes[0] + 0;                     /* Ball(pos, vel, radius) */
        sc.obj[3].p[1] = 4.6; sc.obj[3].p[2] = -0.9; sc.obj[3].p[3] = 0.9; sc.obj[3].p[4] = 1.0; sc.obj[3].p[8] = 0.35;
        uint64_t id = 0;
        if (compile(ctx, joined, 0, &sc /* built for THIS scene's metric variant */, &id) || id == 0) { fprintf(stderr, "rtgr_user_unit_compile: %s\n", err()); return 22; }
        sc.user_metric = id;
        rtgr_unit_info ui;
        if (info(ctx, id, &ui) || ui.metric != RTGR_KS_REF || ui.spin != 0 || !ui.has_objects || !ui.has_reach || !ui.probe_ok) {
            fprintf(stderr, "rtgr_user_unit_info: metric %u spin %u objects %u reach %u probe %u (%s)\n", ui.metric, ui.spin, ui.has_objects, ui.has_reach, ui.probe_ok, err());
            return 23;
        }
        rtgr_camera cam;
        memset(&cam, 0, sizeof cam);
        cam.pos[1] = 4.0; cam.pos[2] = -2.0; cam.widthx[1] = 1.0; cam.widthy[3] = 1.0; cam.normal[2] = 1.0;
        rtgr_solver opt;
        if (defaults(&opt, 0)) return 5;
        if (check(ctx, &sc, &opt, &cam, 48, 48, 0)) { fprintf(stderr, "rtgr_scene_check: %s\n", err()); return 24; }
        const uint64_t ni = 64, nj = 64, n = ni * nj;
        double* rgb = (double*)calloc(3 * n, sizeof(double));
        uint8_t* hit = (uint8_t*)calloc(n, 1);
        uint8_t* status = (uint8_t*)calloc(n, 1);
        uint32_t* nacc = (uint32_t*)calloc(n, 4);
        uint32_t* nrej = (uint32_t*)calloc(n, 4);
        rtgr_ray_outputs outs;
        memset(&outs, 0, sizeof outs);
        outs.hit = hit; outs.status = status; outs.n_accept = nacc; outs.n_reject = nrej;
        rtgr_counters ctr;
        if (trace(ctx, &sc, &opt, NULL, &cam, ni, nj, 0, nj, rgb, &outs, &ctr)) { fprintf(stderr, "rtgr_trace_f64: %s\n", err()); return 7; }
        if (ctr.rays != n) return 8;
        sc.user_metric = 0;                  /* a scene with user objects and no unit is refused, not traced with something else */
        if (trace(ctx, &sc, &opt, NULL, &cam, ni, nj, 0, nj, rgb, NULL, NULL) == 0) { fprintf(stderr, "a scene without its unit was traced\n"); return 25; }
        FILE* f = fopen(argv[3], "wb");
        if (!f) return 9;
        fwrite(rgb, sizeof(double), 3 * n, f);
        fwrite(hit, 1, n, f);
        fwrite(status, 1, n, f);
        fwrite(nacc, 4, n, f);
        fwrite(nrej, 4, n, f);
        fclose(f);
        printf("ok %llu rays, unit %llx: far %u near %u f32 %u waves/SIMD\n", (unsigned long long)ctr.rays, (unsigned long long)id, ui.far_waves, ui.near_waves, ui.f32_waves);
        return 0;
    }
    /* example2(): src/RayTraceGR.jl:581-593 */
    rtgr_scene sc;
    memset(&sc, 0, sizeof sc);
    sc.metric = RTGR_KS_REF; sc.M = 1.0; sc.a = 0.0; sc.nobj = 3;
    sc.obj[0].kind = RTGR_SPHERE; sc.obj[0].p[4] = 1.0; sc.obj[0].p[8] = -10.0;               /* caelum */
    sc.obj[1].kind = RTGR_PLANE; sc.obj[1].p[0] = -20.0;                                       /* frustum */
    sc.obj[2].kind = RTGR_SPHERE; sc.obj[2].p[1] = 4.0; sc.obj[2].p[4] = 1.0; sc.obj[2].p[8] = 0.5;  /* sphere */
    rtgr_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.pos[1] = 4.0; cam.pos[2] = -2.0; cam.widthx[1] = 1.0; cam.widthy[3] = 1.0; cam.normal[2] = 1.0;
    rtgr_solver opt;
    if (defaults(&opt, 0)) return 5;
    const uint64_t ni = 200, nj = 200, n = ni * nj;
    double* st = (double*)malloc(n * 8 * sizeof(double));
    pixel_f64* px = (pixel_f64*)calloc(n, sizeof(pixel_f64));
    pixel_f64* out = (pixel_f64*)calloc(n, sizeof(pixel_f64));
    if (canvas(ctx, &sc, &cam, ni, nj, 0, nj, st)) { fprintf(stderr, "make_canvas: %s\n", err()); return 6; }
    for (uint64_t k = 0; k < n; k++) {   /* Pixel(pos, normal, zeros)  (:475) at pixels[i,j], linear index i + j*ni */
        memcpy(px[k].pos, st + 8 * k, 32);
        memcpy(px[k].normal, st + 8 * k + 4, 32);
    }
    rtgr_counters ctr;
    if (pixels(ctx, &sc, &opt, (const double*)px, ni, nj, (double*)out, &ctr)) { fprintf(stderr, "trace_pixels: %s\n", err()); return 7; }
    if (ctr.rays != n || ctr.events != n) { fprintf(stderr, "counters: %llu rays %llu events\n", (unsigned long long)ctr.rays, (unsigned long long)ctr.events); return 8; }
    FILE* f = fopen(argv[3], "wb");
    if (!f) return 9;
    for (uint64_t k = 0; k < n; k++) {   /* image[j][i][c]: k = i + j*ni is already row j, column i */
        if (memcmp(out[k].pos, px[k].pos, 64) != 0) { fprintf(stderr, "pos/normal not preserved at %llu\n", (unsigned long long)k); return 10; }
        for (int c = 0; c < 3; c++) {
            double v = out[k].rgb[c];
            v = v < 0 ? 0 : (v > 1 ? 1 : v);
            fputc((int)nearbyint(v * 255.0), f);
        }
    }
    fclose(f);
    /* legacy trace_ray(metric, objs, cb, p)::Pixel on the centre pixel (i,j) = (100,100), 1-based */
    const uint64_t kc = 99 + 99 * ni;
    double rgb1[3], se[8];
    uint8_t status = 255;
    if (one(ctx, &sc, &opt, px[kc].pos, px[kc].normal, rgb1, se, &status)) { fprintf(stderr, "trace_one: %s\n", err()); return 11; }
    for (int c = 0; c < 3; c++)
        if (rgb1[c] != out[kc].rgb[c]) { fprintf(stderr, "trace_one differs from trace_pixels\n"); return 12; }
    if (status != RTGR_RAY_EVENT) return 13;
    if (ctx && ((fn_destroy)dlsym(h, "rtgr_destroy"))(ctx)) return 16;
    printf("ok %llu rays\n", (unsigned long long)ctr.rays);
    return 0;
}
