// rtgr_args.hpp — plain-old-data shared by host and device code: device-side scene / solver / camera, the argument
// blocks of the pipeline's kernels and the layout of the per-ray records in the workspace.  No device code here, so the
// host-only translation units (rtgr_context.hip …) can include it without instantiating kernels.
#pragma once
#include <stdint.h>

#include "../../include/rtgr.h"

namespace rtgr {

template <class R>
struct DevObject {
    uint32_t kind;
    uint32_t type;   // RTGR_USER_OBJECT: the caller's tag for rtgr_user_distance / rtgr_user_objcolor (rtgr_object.type)
    R p[9];
    uint32_t orig;   // the object's index in the CALLER's list (the device list is regrouped: DevScene)
};

// The object list (src/RayTraceGR.jl:433-441: a Vector of any length).  The first RTGR_MAX_OBJECTS objects sit in the kernels' argument
// block (scalar loads from the kernarg segment), the rest — rare — in a device table the context keeps per distinct list
// (rtgr_context.hip: object_table); `more` is null when nobj <= RTGR_MAX_OBJECTS.  Kernels walk the list with for_each_object /
// for_each_by_kind (rtgr_physics.hpp), never by index.
// ORDER.  The caller's order matters to the colour rule only (first-smaller-wins and the scale omin / length(objs),
// src/RayTraceGR.jl:520-530); the event condition is a minimum (:433-441) and the FAR pass's reach test a conjunction — neither
// cares.  So the device list is REGROUPED by the host: the spheres first (nsph of them, in the caller's order), then everything else
// (in the caller's order), each object carrying its original index (DevObject::orig), which the colour rule breaks ties by and
// reports.  The integrate kernels then walk the spheres — all but two or three objects of any long list — in a loop of their own
// with no dispatch on the kind: a third of the scalar instructions per object.
template <class R>
struct DevScene {
    uint32_t metric;
    uint32_t nobj;
    uint32_t nsph;   // objects [0, nsph) of the regrouped list are RTGR_SPHEREs
    uint32_t ngroups;   // > 0: the spheres [nloose, nsph) are laid out group by group (see below); 0: no groups
    R M, a;
    DevObject<R> obj[RTGR_MAX_OBJECTS];
    const DevObject<R>* more;   // objects RTGR_MAX_OBJECTS .. nobj-1 (the table holds the WHOLE list: table = more - RTGR_MAX_OBJECTS)
    uint32_t nloose;    // (ngroups > 0) spheres [0, nloose) belong to no group
    uint32_t nsuper;    // > 0: the groups themselves come in that many runs of neighbouring groups with a bounding sphere each (a second level)
};
// GROUPS (long lists: DESIGN.md §4.7).  The FAR pass's reach test asks of every object, every step, "can this step reach you?"; of a
// list of 64 small spheres the answer is no for all but one or two.  The host therefore sorts the spheres of a long list into groups of
// up to RTGR_GROUP_MAX neighbours (median splits of their centres: rtgr_context.hip) and gives every group a bounding sphere; a step
// that provably stays outside a group's bounding sphere cannot change the sign of any member's distance, and the members are only
// asked when some lane of the wave cannot prove that.  A group is a DevObject in sphere form — p[1..3] the centre, p[8] the radius,
// `type` the position of its first member in the device list, `orig` their number — and the groups follow the objects in the table:
// groups = table + nobj.  Lists of a few hundred spheres get a second level the same way: runs of ~8 neighbouring groups (a subtree of the
// same median splits) with a bounding sphere over all their members, asked before their groups — supers = groups + ngroups, `type` the
// first GROUP of the run, `orig` their number.  Same results bit for bit with and without (the bound is a bound; FULL == FAR + NEAR is under test with
// grouped lists); option groups = 0 switches them off (A/B, tests).
#define RTGR_GROUP_MAX 8

template <class R>
struct DevSolver {
    R reltol, abstol, lambda0, lambda1, hit_threshold;
    R miss_rgb[3];
    uint32_t max_steps, interp_points;
};

template <class R>
struct DevCamera {
    R pos[4], widthx[4], widthy[4], normal[4];
};

enum LaneState : int { L_FREE = 0, L_TAKEN = 1, L_RUN = 2, L_EXIT = 3 };

// event record layout (scalars of type R per ray)
constexpr int REC_X = 0;      // x[4]   position at the start of the last step
constexpr int REC_C = 4;      // c[m][q], m = 0..3 (θ¹..θ⁴), q = 0..3
constexpr int REC_PS = 20;    // sign of the callback condition at the step start (0: no event, use θ = top = 0)
constexpr int REC_TOP = 21;   // bracket top θ
constexpr int REC_T = 22;     // λ at the step start
constexpr int REC_H = 23;     // step size
constexpr int REC_U = 24;     // u[4] and cu[m][q] (only when the caller wants state_end)
constexpr int REC_CU = 28;
// widths of the record proper: 24 scalars, 44 with the velocity polynomial (how they are stored: RecRef below)
constexpr int REC_W = 24;
constexpr int REC_W_STATE = 44;

template <class R>
struct TraceArgs {
    DevScene<R> sc;
    DevSolver<R> opt;
    DevCamera<R> cam;
    const R* state0;  // n x 8 or null (camera)
    uint64_t ni, nj, j0, nrows;
    uint64_t jstride; // local row k is image row j0 + k*jstride (1 = contiguous slab; N = cyclic rows of an N-way split)
    R* rgb;           // 3 planes of n
    R* state_end;     // optional
    R* lambda_end;
    uint8_t* status;
    uint8_t* hit;
    uint32_t* hit32;  // the same as 32 bits (object lists beyond 255)
    uint32_t* n_accept;
    uint32_t* n_reject;
    unsigned long long* counters;  // rtgr_counters or null
    // the outputs may be a window of larger arrays (host entry points pipeline a job piece by piece, the multi-device path
    // writes a rank's rows): ray w of this call goes to element out_offset + w, rgb planes are plane_stride apart
    uint64_t plane_stride;  // 0: ni * nrows
    uint64_t out_offset;
    uint32_t* nan_flag;     // optional: set to 1 when a caller-supplied state0 holds a NaN (the reference asserts, :279)
};

template <class R>
struct IntegrateArgs {
    DevScene<R> sc;
    DevSolver<R> opt;
    const R* state0;        // n x 8
    const uint32_t* order;  // queue position -> ray index (longest-expected-first), or null = natural order
    uint64_t n;             // rays in this chunk
    R* rec;                 // n x recw: the event records' TAILS (RecRef)
    uint32_t* meta;         // n x 3: accepted, rejected, status | interior << 8
    int recw;               // REC_TAIL or REC_TAIL_STATE
    R* hand;                // n x HAND_W: start / hand-over records; an ending ray's event record overlays its own (RecRef)
    unsigned long long* ctrl;  // [0] ray queue head of the FULL / FAR pass, [1] queue head of the NEAR pass (per round)
    uint32_t pick_flag;     // passes that resume rays (NEAR, FAR of round >= 1): meta flag of the rays to pick up; 0 = camera rays
    uint32_t allow_handback;  // NEAR: hand a ray back to the next round's FAR pass once it has left every object's reach
    uint32_t handback_after;  // … and has stayed this many accepted steps in the NEAR pass (0: any stay; round 6's "long stayers only")
    uint32_t queue_chunk;   // ray ids popped per atomic: <= RTGR_QUEUE_CHUNK, smaller when a wave gets few rays in total
    unsigned long long* counters;
    // prepare_kernel only: where the rays come from (state0 == null: the camera) and the ordering key outputs
    DevCamera<R> cam;
    uint64_t ni, nj, j0, jstride, first;  // ray w of the chunk is pixel idx = first + w: i = idx % ni, j = j0 + (idx / ni) * jstride
    uint8_t* keys;          // n ordering keys (or null: natural order)
    uint32_t* hist;         // 256-bin histogram of the keys
    // FAR -> NEAR: ids of the rays handed over with fewer than near_early accepted steps (appended with ctrl[6] as the
    // cursor); the NEAR pass starts with those
    uint32_t* early;
    uint32_t near_early;
    uint32_t n_simd;        // SIMDs of the device (4 per CU): workgroup b is the (b / n_simd)-th oldest wave of its SIMD
    uint32_t fair_shift;    // != 0: the waves of a SIMD take turns at the top priority, slices of 2^fair_shift clocks
    uint32_t* nan_flag;     // prepare_kernel: raised when a caller-supplied ray state holds a NaN (:279), or null
#ifdef RTGR_ROOT_STATS
    unsigned long long* dbg;  // debug builds: per-wave {start, end, iterations, rays} of the NEAR pass, then per-ray stays
#endif
};

// Integrate passes.  FULL: every accepted step runs the ContinuousCallback scan (8 interior samples + end point).
// FAR / NEAR split the same work by phase of the ray: the FAR pass replaces the scan by a rigorous per-object bound
// ("no object's distance can change sign anywhere in this step"); a ray for which the bound fails is handed — with its
// PRE-step state, so the step is simply redone — to the NEAR pass, which is the FULL algorithm started from a
// hand-over record instead of a camera ray.  Results are identical to FULL by construction (the scan is skipped only
// where it provably finds nothing); rays spend >90 % of their steps in the FAR pass, which is ~30 % cheaper per step.
enum IntegrateMode : int { MODE_FULL = 0, MODE_FAR = 1, MODE_NEAR = 2 };
// A ray's start / hand-over record and its event record are never alive at the same time (prepare -> FAR -> [hand-over ->] NEAR
// -> event record -> resolve: each writer holds the ray in registers when it writes), so the event record's first HAND_W scalars
// OVERLAY the hand-over record (array `hand`: one 128-byte line per ray in Float64, always aligned) and only its tail — 8 scalars,
// 28 with the velocity polynomial — has storage of its own (array `rec`, stride recw = REC_TAIL / REC_TAIL_STATE).  Round 3 kept
// the two records apart (128 B per ray more).  Two layouts tried and measured on the way (profiles/r04/README.md): one slot of 24
// scalars per ray made every other hand-over record straddle a line (HBM traffic 991 -> 1124 B per ray), one slot padded to 32 made
// resolve fetch two full lines per ray (1089).
constexpr int HAND_W = 16;  // x[4] u[4] k0[4] t dt ps lq
constexpr int REC_TAIL = REC_W - HAND_W;              // 8
constexpr int REC_TAIL_STATE = REC_W_STATE - HAND_W;  // 28
template <class P>
struct RecRef {   // rec[i] of the event record of one ray: i < HAND_W lives in the ray's hand-over line, the rest in its tail
    P* head; P* tail;
    __host__ __device__ P& operator[](int i) const { return i < HAND_W ? head[i] : tail[i - HAND_W]; }
};
constexpr uint32_t META_HANDED = 0xffff0000u;    // meta[3*idx+2] of a ray waiting for a NEAR pass
constexpr uint32_t META_HANDBACK = 0xffff0001u;  // … of a ray a NEAR pass handed back to the next round's FAR pass
constexpr uint32_t META_HANDED_EARLY = 0xffff0002u;  // … of a ray waiting for the NEAR pass ON ITS EARLY LIST
#ifndef RTGR_QUEUE_CHUNK
#define RTGR_QUEUE_CHUNK 256ull
#endif

template <class R>
struct ResolveArgs {
    DevScene<R> sc;
    DevSolver<R> opt;
    const R* rec;      // event-record tails (stride recw) …
    const R* hand;     // … and heads (the rays' hand-over lines, stride HAND_W)
    const uint32_t* meta;
    int recw;
    uint32_t select;   // 1: lists beyond the argument block are narrowed per wave to the objects a ray's last step can meet (select_objects)
    uint64_t n;        // rays in this chunk
    uint64_t offset;   // first ray of the chunk in the caller's slab
    uint64_t n_slab;   // rays in the slab (plane stride of rgb)
    R* rgb;
    R* state_end;
    R* lambda_end;
    uint8_t* status;
    uint8_t* hit;
    uint32_t* hit32;
    uint32_t* n_accept;
    uint32_t* n_reject;
};

}  // namespace rtgr
