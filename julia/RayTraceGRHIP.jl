# RayTraceGRHIP.jl — the reference-side binding a RayTraceGR.jl maintainer would add to route the hot path
# (`trace_rays`, src/RayTraceGR.jl:482-536) through librtgr_hip.so (include/rtgr.h, ABI version 4).
#
# NOT EXECUTED IN THIS REPOSITORY: the build image has no Julia.  What stands in for running it:
#   * tests/c/abi_layout.c — a compiled C caller that passes the same bytes this file would (structs by pointer, an
#     88-byte Pixel array) and whose _Static_asserts pin the table below; tests/test_abi.py runs it on the CPU (layout,
#     symbols) and on the GPU (example2() == sphere2.png through rtgr_trace_pixels_f64 + rtgr_trace_one_f64; a
#     KerrSchild(1, 0.998) + Disk frame through rtgr_trace_f64 with per-ray outputs, against the oracle);
#   * raytracegr.jl_amd/api.py — the same calls through Python ctypes, which every GPU test uses;
#   * tests/test_julia_stub.py — every `ccall` below against the prototypes of include/rtgr.h (symbol declared and exported,
#     as many argument types and arguments as C parameters, pointer / Cint / UInt64 kinds, return type), the structs' field
#     order and the enum constants against the header's, the block structure of this file and of julia/runtests_hip.jl, and
#     that every name runtests_hip.jl uses is defined here.
#   * julia/runtests_hip.jl — the reference's test/runtests.jl:12-79 (incl. the commented-out "rays" test) restated over
#     this module, plus example1() / example2() against the committed goldens: what a maintainer runs first.
#
# fieldoffset table (bytes; Julia lays isbits structs out by the C rules, so `fieldoffset(T, i)` must print exactly this —
# a maintainer can check with `[(fieldname(T,i), fieldoffset(T,i)) for i in 1:fieldcount(T)]`; runtests_hip.jl does):
#
#   RtgrObject       80   kind 0, type 4, p 8
#   RtgrScene      1320   metric 0, nobj 4, M 8, a 16, user_metric 24, obj 32, objects 1312
#   RtgrSolver       72   reltol 0, abstol 8, lambda0 16, lambda1 24, hit_threshold 32, miss_rgb 40, max_steps 64, interp_points 68
#   RtgrCamera      128   pos 0, widthx 32, widthy 64, normal 96
#   RtgrCounters     64   rays 0, accepted 8, rejected 16, rhs_evals 24, events 32, events_interior 40, not_finished 48, reserved 56
#   RtgrRayOutputs   64   state_end 0, lambda_end 8, status 16, hit 24, n_accept 32, n_reject 40, redshift 48, hit32 56
#   Pixel{Float64}   88   pos 0, normal 32, rgb 64          (the reference's own type, src/RayTraceGR.jl:446-450)
#   Pixel{Float32}   44   pos 0, normal 16, rgb 32
#
# It is a thin `ccall` layer; host code stays Julia, the metric/object/Pixel signatures of the reference are preserved,
# and anything that cannot cross the C ABI (an arbitrary metric callable, an Object subtype without device source, a scalar
# type other than Float64 / Float32) falls back to the reference's own CPU path — never silently: every such return goes
# through `cpu_fallback`, which emits one `@warn` naming the reason (a 3 ms frame and a 10 s one must not look alike).
#
# Every BASELINE.json configuration from Julia (INTEGRATION.md "Julia" has them spelled out):
#   C1  example1()                                                      minkowski, 200², == scenes/sphere.png
#   C2  example2(ni = 1024, nj = 1024, metric = KerrSchild(1.0, 0.8))   Kerr–Schild a = 0.8, 3 objects
#   C3  render(kerr_schild, objs, cam..., 4096, 4096; ctx = Context(0:7))   rows dealt to 8 GPUs, RGB planes back
#   C4  example2(T = Float32, ni = 2048, nj = 2048, metric = KerrSchild(1.0, 0.8))
#   C5  example_disk(ni = 8192, nj = 8192)                              KerrSchild(1, 0.998) + Disk(0.05, 2, 4)
module RayTraceGRHIP

using RayTraceGR
using StaticArrays

const librtgr = get(ENV, "RTGR_LIB", "librtgr_hip.so")
const RTGR_MAX_OBJECTS = 16      # objects held inline in RtgrScene.obj; a longer `objs` goes through RtgrScene.objects (any length)
const Ctx = Ptr{Cvoid}          # rtgr_context*; C_NULL = the process's default context
const D = RayTraceGR.D          # 4 (src/RayTraceGR.jl:253-254)

# ---- PODs of include/rtgr.h ------------------------------------------------------------------------------------------
struct RtgrObject
    kind::UInt32
    type::UInt32                # RTGR_USER_OBJECT: the tag handed to the unit's rtgr_user_distance / rtgr_user_objcolor
    p::NTuple{9,Float64}
end
struct RtgrScene
    metric::UInt32
    nobj::UInt32
    M::Float64
    a::Float64
    user_metric::UInt64
    obj::NTuple{RTGR_MAX_OBJECTS,RtgrObject}
    objects::Ptr{RtgrObject}    # C_NULL: the list is obj[1:nobj]; otherwise the WHOLE list, nobj objects — `objs` of any length (:433-441)
end
struct RtgrSolver
    reltol::Float64
    abstol::Float64
    lambda0::Float64
    lambda1::Float64
    hit_threshold::Float64
    miss_rgb::NTuple{3,Float64}
    max_steps::UInt32
    interp_points::UInt32
end
struct RtgrCamera
    pos::NTuple{4,Float64}
    widthx::NTuple{4,Float64}
    widthy::NTuple{4,Float64}
    normal::NTuple{4,Float64}
end
struct RtgrCounters
    rays::UInt64; accepted::UInt64; rejected::UInt64; rhs_evals::UInt64
    events::UInt64; events_interior::UInt64; not_finished::UInt64; reserved::UInt64
end
struct RtgrRayOutputs          # optional per-ray outputs; C_NULL = not wanted
    state_end::Ptr{Cvoid}
    lambda_end::Ptr{Cvoid}
    status::Ptr{UInt8}
    hit::Ptr{UInt8}
    n_accept::Ptr{UInt32}
    n_reject::Ptr{UInt32}
    redshift::Ptr{Cvoid}
    hit32::Ptr{UInt32}          # omin as 32 bits: object lists beyond 255
end

# enum rtgr_metric / rtgr_object_kind / rtgr_ray_status of the header (tests/test_julia_stub.py compares the values)
const RTGR_MINKOWSKI = UInt32(0)
const RTGR_KS_REF = UInt32(1)
const RTGR_KS_TRUE = UInt32(2)
const RTGR_USER = UInt32(3)
const RTGR_METRIC_GENERIC = UInt32(0x100)
const RTGR_PLANE = UInt32(1)
const RTGR_SPHERE = UInt32(2)
const RTGR_DISK = UInt32(3)
const RTGR_USER_OBJECT = UInt32(4)
const RTGR_RAY_EVENT = UInt8(0)
const RTGR_RAY_LAMBDA1 = UInt8(1)
const RTGR_RAY_MAXSTEPS = UInt8(2)
const RTGR_RAY_DTMIN = UInt8(3)
const RTGR_RAY_NAN = UInt8(4)

function check(rc)
    rc < 0 && error("librtgr_hip: ", unsafe_string(ccall((:rtgr_last_error, librtgr), Cstring, ())))
    rc
end

"""
    cpu_fallback(what, why)

Every place where a legal call of the reference cannot cross the C ABI and is handed to the reference's own CPU path goes through
here: ONE `@warn` per distinct (entry point, reason) — the CPU path is ~3000 x slower than the device (profiles/r05/cpu_baseline_1024.json),
and a first user must be able to tell the two apart by something better than the clock.  `ENV["RTGR_QUIET_FALLBACK"] = "1"` silences it.
"""
function cpu_fallback(what::AbstractString, why::AbstractString)
    get(ENV, "RTGR_QUIET_FALLBACK", "0") == "1" ||
        @warn "RayTraceGRHIP.$what: $why — running the reference's CPU path (RayTraceGR.$what), not the GPU" maxlog = 1 _id = Symbol(what, hash(why))
    nothing
end

# ---- new scene vocabulary (SURVEY §8 f4): what BASELINE configs 2, 4 and 5 need and the reference does not have -------
"""
    KerrSchild(M, a; textbook = true, generic = false)

The parameterised Kerr–Schild metric the reference's `kerr_schild` hard-wires to `M = 1`, `a = 0  # T(0.8)`
(src/RayTraceGR.jl:275-276).  A callable like every metric of the reference — `KerrSchild(1.0, 0.8)(x)` is the 4x4 `g_ab`
in plain Julia, accepts `Dual`s, works with `RayTraceGR.dmetric / christoffel / make_canvas / trace_rays` on the CPU — and
the C ABI carries it as an enum plus `(M, a)`:
`textbook = true`  → `RTGR_KS_TRUE`, the radius `r² = (q + sqrt(q² + 4a²z²))/2`, `q = ρ² − a²` (a true Kerr black hole);
`textbook = false` → `RTGR_KS_REF`, the radius exactly as written at :284 (not a solution of the field equations for a ≠ 0;
identical for a = 0).  `generic = true` traces with the reference's own formulation of the RHS (4-wide duals, 4x4 inverse,
Christoffel contraction) instead of the closed Kerr–Schild contraction: same results to rounding, ~2.6 x the time.
"""
struct KerrSchild
    M::Float64
    a::Float64
    textbook::Bool
    generic::Bool
end
KerrSchild(M::Real = 1.0, a::Real = 0.0; textbook::Bool = true, generic::Bool = false) =
    KerrSchild(Float64(M), Float64(a), textbook, generic)

basereal(::Type{T}) where {T<:AbstractFloat} = T
basereal(::Type{RayTraceGR.Dual{T,DT}}) where {T,DT} = T             # the reference's Dual mixes with its OWN scalar type only (:55-121)
function (m::KerrSchild)(xx::SVector{4,T}) where {T}             # src/RayTraceGR.jl:274-294 with (M, a) as parameters
    M, a = basereal(T)(m.M), basereal(T)(m.a)
    t, x, y, z = xx
    @assert !any(isnan, (t, x, y, z))                             # :279
    η = @SMatrix T[p == q ? (p == 1 ? -1 : 1) : 0 for p in 1:4, q in 1:4]
    ρ2 = x^2 + y^2 + z^2
    if m.textbook
        q = ρ2 - a^2
        r = sqrt((q + sqrt(q^2 + 4 * a^2 * z^2)) / 2)
    else
        r = sqrt(ρ2 - a^2) / 2 + sqrt(a^2 * z^2 + ((ρ2 - a^2) / 2)^2)   # :284 as written
    end
    f = 2 * M * r^3 / (r^4 + a^2 * z^2)                           # :285
    k = SVector{4,T}(1, (r * x + a * y) / (r^2 + a^2), (r * y - a * x) / (r^2 + a^2), z / r)   # :286-289
    @SMatrix T[η[p, q] + f * k[p] * k[q] for p in 1:4, q in 1:4]   # :291
end

"""
    Disk{T}(half_thickness, r_in, r_out) <: RayTraceGR.Object{T}

Thin accretion disk `|z| <= h`, `r_in <= sqrt(x² + y²) <= r_out` in the equatorial plane (BASELINE config 5; no reference
counterpart).  Obeys the reference's distance contract (:377-383: zero on the surface, positive outside, negative inside),
so the reference's CPU `trace_rays` handles it too through the two methods below; `RTGR_DISK` across the ABI.
"""
struct Disk{T} <: RayTraceGR.Object{T}
    half_thickness::T
    r_in::T
    r_out::T
end
function RayTraceGR.distance(d::Disk{T}, pos::SVector{4,T})::T where {T}
    ϱ = sqrt(pos[2]^2 + pos[3]^2)
    max(abs(pos[4]) - d.half_thickness, d.r_in - ϱ, ϱ - d.r_out)
end
function RayTraceGR.objcolor(d::Disk{T}, pos::SVector{4,T})::SVector{3,T} where {T}   # checkerboard in (radius, azimuth)
    ϱ = sqrt(pos[2]^2 + pos[3]^2)
    ϕ = atan(pos[3], pos[2])
    SVector{3,T}(1, mod(ϱ, 1), mod(12 * ϕ / π, 1))
end

# the objects' own parameters stay Float64 across the ABI (rtgr_object.p); T selects the arithmetic of the path
pack(pl::RayTraceGR.Plane) = RtgrObject(RTGR_PLANE, 0, (Float64(pl.time), 0, 0, 0, 0, 0, 0, 0, 0))
pack(s::RayTraceGR.Sphere) = RtgrObject(RTGR_SPHERE, 0, (Float64.(s.pos)..., Float64.(s.vel)..., Float64(s.radius)))
pack(d::Disk) = RtgrObject(RTGR_DISK, 0, (Float64(d.half_thickness), Float64(d.r_in), Float64(d.r_out), 0, 0, 0, 0, 0, 0))
"""
    DeviceObjects(source)                       a family of NEW Object subtypes, given as device source
    DeviceObject{T}(family, type, fields...)    one object of it: `<: RayTraceGR.Object{T}`, goes into `objs` like a Sphere

The reference's `Object{T}` is an open abstract type (src/RayTraceGR.jl:374-389): a new subtype brings `distance(obj, pos)` and
`objcolor(obj, pos)`.  A Julia method cannot cross the C ABI; the native counterpart is the same two methods as C++ source,

    template <class S> __device__ S    rtgr_user_distance(unsigned type, const S x[4], const S p[9]);
    template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]);

(optionally `rtgr_user_reach`: include/rtgr.h "user objects"), compiled at run time — together with the metric the objects are
traced with — into the scene's unit (`rtgr_user_unit_compile`: one ccall, built in-process, no hipcc).  `type` tells the family's
object types apart inside the source, `fields` (up to 9 numbers) are the object's parameters `p`.  DeviceObjects of SEVERAL families
may stand in one `objs` vector when each family says how many types its source defines (`DeviceObjects(source, ntypes = 2)`): their
sources are joined into one (`rtgr_user_source_join`: a namespace per family, type tags renumbered family after family in the order
of first appearance).  An object type WITHOUT device source still takes the reference's CPU path, as before.
"""
struct DeviceObjects
    source::String
    ntypes::UInt32          # how many object types the source defines (0: not stated — enough as long as a scene uses one family)
end
DeviceObjects(source::AbstractString; ntypes::Integer = 0) = DeviceObjects(String(source), UInt32(ntypes))
struct DeviceObject{T} <: RayTraceGR.Object{T}
    family::DeviceObjects
    type::UInt32
    p::NTuple{9,Float64}
end
function DeviceObject{T}(family::DeviceObjects, type::Integer, fields::Real...) where {T}
    length(fields) <= 9 || error("an object has at most 9 scalar fields (rtgr_object.p)")
    DeviceObject{T}(family, UInt32(type), ntuple(i -> i <= length(fields) ? Float64(fields[i]) : 0.0, 9))
end
# (the two methods exist on the device only: the reference's CPU trace_rays cannot evaluate them — say so instead of a MethodError)
RayTraceGR.distance(o::DeviceObject{T}, pos::SVector{4,T}) where {T} =
    error("a DeviceObject carries device source only; give the type Julia methods `distance` / `objcolor` to trace it on the CPU")
RayTraceGR.objcolor(o::DeviceObject{T}, pos::SVector{4,T}) where {T} =
    error("a DeviceObject carries device source only; give the type Julia methods `distance` / `objcolor` to trace it on the CPU")
pack(o::DeviceObject) = RtgrObject(RTGR_USER_OBJECT, o.type, o.p)
pack(o::RayTraceGR.Object) = nothing            # an object type without device source: the CPU path handles it
const NOOBJ = RtgrObject(0, 0, ntuple(_ -> 0.0, 9))

"""
    Context(device_ids) / close(ctx)

All devices one Julia process drives (`rtgr_create`).  `trace_rays(...; ctx)` on a context with several devices deals the
canvas rows cyclically to ALL of them inside the one `ccall` (`rtgr_trace_pixels_f64/_f32`: device k of N takes rows
k, k+N, …, uploads those rows of `c.pixels` over its own PCIe link and writes them straight back into the output array —
nothing is routed through the first device; include/rtgr.h "the hot path, host buffers") — no Distributed.jl, which the
reference tried and abandoned (README.md:129-135).  Without a context the library's default one (ONE device) is used.
Exercised in this repository by the C caller (`tests/c/abi_layout.c --render … 3`) and `tests/test_gpu_context.py`
with a context that lists the GPU several times; between physically different GPUs it has not run yet (no node).
"""
mutable struct Context
    handle::Ctx
    function Context(device_ids::AbstractVector{<:Integer} = Int[])
        h = Ref{Ctx}(C_NULL)
        ids = collect(Cint, device_ids)
        check(ccall((:rtgr_create, librtgr), Cint, (Ptr{Cint}, Cint, Ptr{Ctx}), isempty(ids) ? C_NULL : ids, length(ids), h))
        finalizer(close, new(h[]))
    end
end
function Base.close(c::Context)     # idempotent; also the finalizer
    c.handle == C_NULL && return nothing
    ccall((:rtgr_destroy, librtgr), Cint, (Ctx,), c.handle)
    c.handle = C_NULL
    nothing
end
handle(::Nothing) = C_NULL
handle(c::Context) = c.handle
ndevices(ctx) = Int(check(ccall((:rtgr_context_devices, librtgr), Cint, (Ctx,), handle(ctx))))

"""
    DeviceMetric(code_object; M = 1.0, a = 0.0)

A metric function of the user's own — the native stand-in for passing a new Julia function as `metric` (:302-309) —
given as C++ source text (`DeviceMetric(source = "...")`: built in-process by `rtgr_user_metric_compile`, no hipcc needed) or
as a gfx950 code object built from `rtgr_user_unit.hip.in` (INTEGRATION.md "A new metric").
Several may be resident at once; a scene names its own by id.
"""
struct DeviceMetric
    code_object::String     # path of a code object (rtgr_user_metric_build, or `python -m raytracegr.jl_amd.user_metric`), or "" when `source` is given
    source::String          # C++ source of rtgr_user_metric<S>: built in-process by the library, one ccall
    stationary::Bool
    M::Float64
    a::Float64
end
DeviceMetric(path::AbstractString; M = 1.0, a = 0.0) = DeviceMetric(path, "", false, M, a)
DeviceMetric(; source::AbstractString, stationary = false, M = 1.0, a = 0.0) = DeviceMetric("", source, stationary, M, a)
"""
    build_metric(source, path; stationary = false) -> DeviceMetric(path)

The build step of `DeviceMetric(source = ...)` on its own (`rtgr_user_metric_build`): source text -> code object file, built by the
library in-process — no GPU, no hipcc needed —, to be kept and loaded in later sessions with `DeviceMetric(path)`.
"""
function build_metric(source::AbstractString, path::AbstractString; stationary = false, M = 1.0, a = 0.0)
    check(ccall((:rtgr_user_metric_build, librtgr), Cint, (Cstring, Cint, Cstring), source, stationary, path))
    DeviceMetric(path, "", stationary, M, a)
end

function module_id(m::DeviceMetric, ctx)
    id = Ref{UInt64}(0)
    if isempty(m.source)
        check(ccall((:rtgr_user_metric_load, librtgr), Cint, (Ctx, Cstring, Ptr{UInt64}), handle(ctx), m.code_object, id))
    else
        check(ccall((:rtgr_user_metric_compile, librtgr), Cint, (Ctx, Cstring, Cint, Ptr{UInt64}),
                    handle(ctx), m.source, m.stationary, id))
    end
    id[]
end

# (enum, M, a, user_metric id) of a metric argument, or nothing when it cannot cross the ABI
metric_desc(m::DeviceMetric, ctx) = (RTGR_USER, m.M, m.a, module_id(m, ctx))
metric_desc(m::KerrSchild, ctx) = ((m.textbook ? RTGR_KS_TRUE : RTGR_KS_REF) | (m.generic ? RTGR_METRIC_GENERIC : UInt32(0)),
                                   m.M, m.a, UInt64(0))
metric_desc(m, ctx) = m === RayTraceGR.minkowski ? (RTGR_MINKOWSKI, 1.0, 0.0, UInt64(0)) :
                      m === RayTraceGR.kerr_schild ? (RTGR_KS_REF, 1.0, 0.0, UInt64(0)) :   # as written: M = 1, a = 0 (:275-276)
                      nothing

# id of the unit that carries the kernels of `family`'s objects for the metric variant of `scene` (rtgr_user_unit_compile reads the
# metric enum, the RTGR_METRIC_GENERIC flag and whether a != 0 off the scene; a DeviceMetric given as source is compiled into the
# same unit) — built once per (context, source, metric variant)
# (no table on this side: rtgr_user_unit_compile answers the same call again — same context, source, `stationary`, metric variant —
#  at once with the id it gave before FOR AS LONG AS THAT UNIT IS RESIDENT, and builds and loads it again after an unload or a
#  refusal by the load-time probe; a Dict here handed out dead ids, ADVICE r5)
function unit_id(family::DeviceObjects, metric, scene, ctx)
    own = metric isa DeviceMetric
    own && isempty(metric.source) && error("DeviceObjects with a DeviceMetric: give the metric as source text too (the two share one unit)")
    source = own ? metric.source * "\n" * family.source : family.source
    id = Ref{UInt64}(0)
    check(ccall((:rtgr_user_unit_compile, librtgr), Cint, (Ctx, Cstring, Cint, Ptr{RtgrScene}, Ptr{UInt64}),
                handle(ctx), source, own && metric.stationary, own ? C_NULL : scene, id))
    id[]
end

# the families of a scene as ONE family (rtgr_user_source_join) and the base each family's type tags move to
const JOINED = Dict{Vector{DeviceObjects},DeviceObjects}()
function join_families(fams::Vector{DeviceObjects})
    all(f -> f.ntypes > 0, fams) ||
        error("DeviceObjects of several families in one scene: say how many object types each source defines — DeviceObjects(source, ntypes = n)")
    bases = UInt32[sum(UInt32[f.ntypes for f in fams[1:k-1]]; init = UInt32(0)) for k in 1:length(fams)]
    joined = get!(JOINED, fams) do
        srcs = Cstring[Base.unsafe_convert(Cstring, f.source) for f in fams]
        nt = UInt32[f.ntypes for f in fams]
        need = Ref{UInt64}(0)
        GC.@preserve fams begin
            check(ccall((:rtgr_user_source_join, librtgr), Cint, (Ptr{Cstring}, Ptr{UInt32}, Cint, Ptr{UInt8}, UInt64, Ptr{UInt64}),
                        srcs, nt, length(fams), C_NULL, 0, need))
            buf = Vector{UInt8}(undef, need[])
            check(ccall((:rtgr_user_source_join, librtgr), Cint, (Ptr{Cstring}, Ptr{UInt32}, Cint, Ptr{UInt8}, UInt64, Ptr{UInt64}),
                        srcs, nt, length(fams), buf, need[], need))
        end
        DeviceObjects(String(buf[1:end-1]), sum(nt))
    end
    joined, bases
end

"""
    Scene

An `rtgr_scene` ready for a `ccall`, together with what its pointers point to: `objs::Vector{Object{T}}` has no length limit in the
reference (src/RayTraceGR.jl:433-441, :483), so a list beyond the 16 inline slots is handed over as an array (`rtgr_scene.objects`)
that must outlive the call.  `ccall(..., (Ptr{RtgrScene}, ...), scene, ...)` roots this object for the duration of the call
(`cconvert` returns it, `unsafe_convert` takes the pointer of the struct inside).
"""
mutable struct Scene
    ref::Base.RefValue{RtgrScene}
    list::Vector{RtgrObject}       # the packed objects when they travel as an array (empty otherwise)
end
Base.cconvert(::Type{Ptr{RtgrScene}}, s::Scene) = s
Base.unsafe_convert(::Type{Ptr{RtgrScene}}, s::Scene) = Base.unsafe_convert(Ptr{RtgrScene}, s.ref)

# (scene, nothing) or (nothing, why it cannot cross the C ABI)
function scene_of(metric, objs, ctx)
    fams = unique(DeviceObjects[o.family for o in objs if o isa DeviceObject])
    family, bases = length(fams) > 1 ? join_families(fams) : (isempty(fams) ? nothing : fams[1], UInt32[0])
    # (a DeviceMetric beside DeviceObjects lives in the objects' unit: its own module is not loaded)
    d = (!isempty(fams) && metric isa DeviceMetric) ? (RTGR_USER, metric.M, metric.a, UInt64(0)) : metric_desc(metric, ctx)
    d === nothing && return nothing, "the metric `$(nameof(typeof(metric)))` is a Julia callable without a device counterpart " *
                                     "(built-ins: minkowski, kerr_schild, KerrSchild(M, a); your own: DeviceMetric(source = ...))"
    po = map(objs) do o
        o isa DeviceObject ? RtgrObject(RTGR_USER_OBJECT, o.type + bases[findfirst(==(o.family), fams)], o.p) : pack(o)
    end
    k = findfirst(isnothing, po)
    k === nothing || return nothing, "objs[$k] is a `$(nameof(typeof(objs[k])))`: an Object subtype without device source " *
                                     "(built-ins: Plane, Sphere, Disk; your own: DeviceObjects(source))"
    list = length(po) > RTGR_MAX_OBJECTS ? RtgrObject[o for o in po] : RtgrObject[]
    packed = ntuple(i -> (isempty(list) && i <= length(po)) ? po[i] : NOOBJ, RTGR_MAX_OBJECTS)
    make(unit) = Scene(Ref(RtgrScene(d[1], length(objs), d[2], d[3], unit, packed, isempty(list) ? Ptr{RtgrObject}(C_NULL) : pointer(list))), list)
    scene = make(d[4])
    isempty(fams) && return scene, nothing
    make(unit_id(family, metric, scene, ctx)), nothing
end
solver_of(::Type{T}) where {T} = begin
    opt = Ref{RtgrSolver}()
    check(ccall((:rtgr_solver_defaults, librtgr), Cint, (Ptr{RtgrSolver}, Cint), opt, T === Float32 ? 1 : 0))
    opt
end
camera_of(pos, widthx, widthy, normal) =
    Ref(RtgrCamera(Tuple(Float64.(pos)), Tuple(Float64.(widthx)), Tuple(Float64.(widthy)), Tuple(Float64.(normal))))

"""
    make_canvas(metric, pos, widthx, widthy, normal, ni, nj; ctx = nothing) -> Canvas{T}

`RayTraceGR.make_canvas` (src/RayTraceGR.jl:457-478) on the device (`rtgr_make_canvas_f64/_f32`: metric at the pixel, `g⁻¹e_t`,
normalisation — one thread per pixel), returned as the reference's own `Canvas{T}` of `Pixel{T}(x, u, zeros)` (:475).
A metric that cannot cross the ABI (or a scalar type other than Float64 / Float32) falls back to the reference's function, with a warning.
"""
function make_canvas(metric, pos::SVector{4,T}, widthx::SVector{4,T}, widthy::SVector{4,T}, normal::SVector{4,T},
                     ni::Int, nj::Int; ctx = nothing) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, RayTraceGR.Object{T}[], ctx)
    if scene === nothing
        cpu_fallback("make_canvas", why)
        return RayTraceGR.make_canvas(metric, pos, widthx, widthy, normal, ni, nj)
    end
    cam = camera_of(pos, widthx, widthy, normal)
    st = Array{T}(undef, 8, ni, nj)                 # n x 8 ray states, pixel index i + j*ni (column-major pixels[i,j], :463-464)
    GC.@preserve st begin
        if T === Float64
            check(ccall((:rtgr_make_canvas_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrCamera}, UInt64, UInt64, UInt64, UInt64, Ptr{Float64}),
                        handle(ctx), scene, cam, ni, nj, 0, nj, pointer(st)))
        else
            check(ccall((:rtgr_make_canvas_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrCamera}, UInt64, UInt64, UInt64, UInt64, Ptr{Float32}),
                        handle(ctx), scene, cam, ni, nj, 0, nj, pointer(st)))
        end
    end
    pixels = Array{RayTraceGR.Pixel{T}}(undef, ni, nj)
    for j in 1:nj, i in 1:ni
        pixels[i, j] = RayTraceGR.Pixel{T}(SVector{4,T}(st[1, i, j], st[2, i, j], st[3, i, j], st[4, i, j]),
                                           SVector{4,T}(st[5, i, j], st[6, i, j], st[7, i, j], st[8, i, j]),
                                           zeros(SVector{3,T}))
    end
    RayTraceGR.Canvas{T}(pixels)
end
# `make_canvas` is generic in T (src/RayTraceGR.jl:457-462); the device computes in Float64 and Float32
function make_canvas(metric, pos::SVector{4,T}, widthx::SVector{4,T}, widthy::SVector{4,T}, normal::SVector{4,T},
                     ni::Int, nj::Int; ctx = nothing) where {T}
    cpu_fallback("make_canvas", "the scalar type $T has no device arithmetic (Float64 and Float32 do)")
    RayTraceGR.make_canvas(metric, pos, widthx, widthy, normal, ni, nj)
end

"""
    trace_rays(metric, objs, c::Canvas{T}; ctx = nothing) -> Canvas{T},   T = Float64 | Float32

Drop-in for `RayTraceGR.trace_rays` (src/RayTraceGR.jl:483-484).  Passes `pointer(c.pixels)` — the reference's own
`Pixel{T}` AoS (:446-450; 88 bytes for Float64, 44 for Float32) — across the ABI; returns a new canvas with `rgb`
filled (:532).  The tolerance is `eps(T)^(3/4)` as in the reference (:485).  `ctx = Context(0:7)`: all eight GPUs of a
node work on the canvas (rows dealt cyclically); the result does not depend on the number of devices, bit for bit.
`metric`: `minkowski`, `kerr_schild`, `KerrSchild(M, a)`, a `DeviceMetric`; anything else runs the reference's CPU path.
`objs`: `Plane`, `Sphere`, `Disk`, `DeviceObject`s, in any number (the reference's `Vector{Object{T}}` has no length limit, :433-441,
and neither has the device: the first 16 travel in the kernels' argument block, a longer list in a device table with its spheres
sorted into groups of neighbours that are asked first — 64 objects cost ≈ 1.8 ×, 256 … 512 ≈ 1.5–1.85 × the 3-object frame); any other `Object` subtype runs the reference's CPU path.  Every fall-back says so once (`@warn`).

Where parity ends: on rays that are CAPTURED with |u^t| ≳ 10⁶ the step sequence follows the rounding noise of the RHS formulation;
the closed-form kernels (`kerr_schild`, `KerrSchild(M, a)`) then take up to 27 % fewer steps than the reference and end such rays with
another status at the step cap (INTEGRATION.md "Where parity ends").  `KerrSchild(M, a; generic = true)` traces with the reference's
own formulation of the RHS (0.82 of SURVEY §8(d)'s flop count executed, 1.18·10¹⁰ steps/s) and is the parity-first setting.
"""
function trace_rays(metric, objs::Vector{RayTraceGR.Object{T}}, c::RayTraceGR.Canvas{T}; ctx = nothing) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, objs, ctx)
    if scene === nothing
        cpu_fallback("trace_rays", why)
        return RayTraceGR.trace_rays(metric, objs, c)
    end
    opt = solver_of(T)
    ni, nj = size(c.pixels)
    out = similar(c.pixels)
    ctr = Ref{RtgrCounters}()
    GC.@preserve c out begin
        if T === Float64
            check(ccall((:rtgr_trace_pixels_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, UInt64, UInt64, Ptr{Cvoid}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, pointer(c.pixels), ni, nj, pointer(out), ctr))
        else
            check(ccall((:rtgr_trace_pixels_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, UInt64, UInt64, Ptr{Cvoid}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, pointer(c.pixels), ni, nj, pointer(out), ctr))
        end
    end
    RayTraceGR.Canvas{T}(out)
end
# `trace_rays` is generic in T (src/RayTraceGR.jl:483-485): any other scalar type is the reference's own business, loudly
function trace_rays(metric, objs::Vector{RayTraceGR.Object{T}}, c::RayTraceGR.Canvas{T}; ctx = nothing) where {T}
    cpu_fallback("trace_rays", "the scalar type $T has no device arithmetic (Float64 and Float32 do)")
    RayTraceGR.trace_rays(metric, objs, c)
end

"""
    RayDetails{T}

What the reference computes per ray and throws away: `sols.u[i]` (`state_end`, :516), `sol.t[end]` (`lambda_end`, :503), the
solver's return code (`status`, ignored at :502-505), `omin` of the colouring rule (`hit`, :518-526), the accepted / rejected
step counts; plus `redshift` (the frequency ratio observed/emitted from `Sphere.vel`, which the reference stores and never
uses, :411) and the call's `RtgrCounters`.  Arrays are `ni x nj` like `c.pixels` (`state_end`: `8 x ni x nj`).
"""
struct RayDetails{T}
    state_end::Array{T,3}
    lambda_end::Matrix{T}
    status::Matrix{UInt8}
    hit::Matrix{UInt32}             # (rtgr_ray_outputs.hit32: `objs` may hold more than 255 objects)
    n_accept::Matrix{UInt32}
    n_reject::Matrix{UInt32}
    redshift::Matrix{T}
    counters::RtgrCounters
end

"""
    eval_objects(metric, objs, xs::Vector{SVector{4,T}}; ctx = nothing) -> (d, dmin, hit, rgb)

The object side of the path at the points `xs`, on the device (`rtgr_eval_objects_f64/_f32`): `d[o, p]` = `distance(objs[o], xs[p])`
(:377-419; a `DeviceObject`: its unit's `rtgr_user_distance`), `dmin[p]` = `min_distance` (:433-441), and the colouring loop of
`trace_rays` (:513-533) as if a ray had ended at `xs[p]`: `hit[p]` = the object's index (0: nothing within the threshold), `rgb[:, p]`.
What `runtests_hip.jl` holds against the Julia methods of the same object, point by point.
"""
function eval_objects(metric, objs::Vector{RayTraceGR.Object{T}}, xs::Vector{SVector{4,T}}; ctx = nothing) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, objs, ctx)
    scene === nothing && error("eval_objects: this scene runs on the reference's CPU path: ", why)
    length(objs) <= 255 || error("eval_objects: `hit` is a byte per point (at most 255 objects)")
    opt = solver_of(T)
    n = length(xs)
    d = Matrix{T}(undef, max(length(objs), 1), n)        # C layout d[p * nobj + o]
    dmin = Vector{T}(undef, n)
    hit = Vector{UInt8}(undef, n)
    rgb = Matrix{T}(undef, 3, n)                         # C layout rgb[3 p + c]
    GC.@preserve xs d dmin hit rgb begin
        if T === Float64
            check(ccall((:rtgr_eval_objects_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}, Ptr{Cvoid}),
                        handle(ctx), scene, opt, pointer(xs), n, pointer(d), pointer(dmin), pointer(hit), pointer(rgb)))
        else
            check(ccall((:rtgr_eval_objects_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}, Ptr{Cvoid}),
                        handle(ctx), scene, opt, pointer(xs), n, pointer(d), pointer(dmin), pointer(hit), pointer(rgb)))
        end
    end
    d, dmin, hit, rgb
end

"""
    check_scene(metric, objs, pos, widthx, widthy, normal; ni = 48, nj = 48, ctx = nothing)

`rtgr_scene_check`: traces a coarse canvas of this camera through the single FULL pass (every accepted step scanned, as the
reference's ContinuousCallback does, :488-490) and through the FAR + NEAR passes, and throws unless the two frames agree (bit for bit
for a built-in metric).  The check to run once on a scene with `DeviceObject`s whose source brings a `rtgr_user_reach` bound: a bound
that is too small loses hits silently, and this is what says so.
"""
function check_scene(metric, objs::Vector{RayTraceGR.Object{T}}, pos, widthx, widthy, normal; ni::Integer = 48, nj::Integer = 48,
                     ctx = nothing) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, objs, ctx)
    scene === nothing && error("check_scene: this scene runs on the reference's CPU path (nothing to check): ", why)
    opt = solver_of(Float64)
    cam = camera_of(pos, widthx, widthy, normal)
    check(ccall((:rtgr_scene_check, librtgr), Cint, (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{RtgrCamera}, UInt64, UInt64, Cint),
                handle(ctx), scene, opt, cam, ni, nj, 0))
    nothing
end

"""
    render(metric, objs, pos, widthx, widthy, normal, ni, nj; T = Float64, ctx = nothing, details = false)
        -> (R, G, B)  or  ((R, G, B), RayDetails)

`make_canvas` + `trace_rays` in ONE call with the camera on the device (`rtgr_trace_f64/_f32`, `state0 = NULL`): the 88-byte
pixels never exist — nothing goes up, three `ni x nj` planes come back (what `colorview(RGB, R', G', B')` consumes, :566-569).
This is the call for the big screens (4096², 8192²: 5.9 GB of pixels each way otherwise).  All devices of `ctx` take part.
"""
function render(metric, objs, pos, widthx, widthy, normal, ni::Integer, nj::Integer;
                T::Type = Float64, ctx = nothing, details::Bool = false)
    scene, why = scene_of(metric, objs, ctx)
    scene === nothing && error("render has no CPU counterpart in the reference (make_canvas + trace_rays are the reference's calls): ", why)
    opt = solver_of(T)
    cam = camera_of(pos, widthx, widthy, normal)
    rgb = Array{T}(undef, ni, nj, 3)                # plane-major: rgb[:, :, c] is plane c
    ctr = Ref{RtgrCounters}()
    det = details ? RayDetails{T}(Array{T}(undef, 8, ni, nj), Matrix{T}(undef, ni, nj), Matrix{UInt8}(undef, ni, nj),
                                  Matrix{UInt32}(undef, ni, nj), Matrix{UInt32}(undef, ni, nj), Matrix{UInt32}(undef, ni, nj),
                                  Matrix{T}(undef, ni, nj), RtgrCounters(0, 0, 0, 0, 0, 0, 0, 0)) : nothing
    GC.@preserve rgb det begin
        outs = details ? Ref(RtgrRayOutputs(pointer(det.state_end), pointer(det.lambda_end), pointer(det.status), C_NULL,
                                            pointer(det.n_accept), pointer(det.n_reject), pointer(det.redshift), pointer(det.hit))) :
                         Ref(RtgrRayOutputs(C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
        if T === Float64
            check(ccall((:rtgr_trace_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Float64}, Ptr{RtgrCamera}, UInt64, UInt64, UInt64, UInt64,
                         Ptr{Float64}, Ptr{RtgrRayOutputs}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, C_NULL, cam, ni, nj, 0, nj, pointer(rgb), outs, ctr))
        else
            check(ccall((:rtgr_trace_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Float32}, Ptr{RtgrCamera}, UInt64, UInt64, UInt64, UInt64,
                         Ptr{Float32}, Ptr{RtgrRayOutputs}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, C_NULL, cam, ni, nj, 0, nj, pointer(rgb), outs, ctr))
        end
    end
    planes = (rgb[:, :, 1], rgb[:, :, 2], rgb[:, :, 3])
    details || return planes
    planes, RayDetails{T}(det.state_end, det.lambda_end, det.status, det.hit, det.n_accept, det.n_reject, det.redshift, ctr[])
end

"""
    render_frames(metric, objs, cams, ni, nj; T = Float64, ctx = nothing) -> Vector{NTuple{3,Matrix{T}}}

SEVERAL frames of one scene in ONE call (`rtgr_trace_frames_f64/_f32`), two in flight inside the library: frame `k` from the camera
`cams[k] = (pos, widthx, widthy, normal)` (rays generated on the device), returned as its three `ni x nj` planes.  An extension — the
reference renders one frame per call (`example1`, `example2`, src/RayTraceGR.jl:560, :596): a render loop's frames end thin (the last
rays of a pass, the last download), and with two in flight the thin end of one overlaps the start of the next (8-11 % per frame at
1024², INTEGRATION.md "Frames in flight").  Each frame is the frame `render` gives for that camera, bit for bit.
"""
function render_frames(metric, objs, cams::AbstractVector, ni::Integer, nj::Integer; T::Type = Float64, ctx = nothing)
    scene, why = scene_of(metric, objs, ctx)
    scene === nothing && error("render_frames has no CPU counterpart in the reference: ", why)
    opt = solver_of(T)
    K = length(cams)
    packed = RtgrCamera[camera_of(c...)[] for c in cams]
    rgb = [Array{T}(undef, ni, nj, 3) for _ in 1:K]      # plane-major, frame by frame
    ptrs = Ptr{T}[pointer(a) for a in rgb]
    ctrs = Vector{RtgrCounters}(undef, K)
    GC.@preserve rgb begin
        if T === Float64
            check(ccall((:rtgr_trace_frames_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, UInt32, Ptr{RtgrCamera}, Ptr{Ptr{Float64}}, UInt64, UInt64, Ptr{Ptr{Float64}},
                         Ptr{RtgrRayOutputs}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, K, packed, C_NULL, ni, nj, ptrs, C_NULL, ctrs))
        else
            check(ccall((:rtgr_trace_frames_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, UInt32, Ptr{RtgrCamera}, Ptr{Ptr{Float32}}, UInt64, UInt64, Ptr{Ptr{Float32}},
                         Ptr{RtgrRayOutputs}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, K, packed, C_NULL, ni, nj, ptrs, C_NULL, ctrs))
        end
    end
    [(a[:, :, 1], a[:, :, 2], a[:, :, 3]) for a in rgb]
end

"""
    trace_rays_frames(metric, objs, canvases::Vector{Canvas{T}}; ctx = nothing) -> Vector{Canvas{T}}

`trace_rays` over several canvases in one call, two in flight (`rtgr_trace_frames_pixels_f64/_f32`): each canvas's own `Pixel{T}`
array goes across the ABI as in `trace_rays`, all of the same size.  Falls back — loudly — to a loop of the reference's `trace_rays`
where the scene cannot cross the ABI.
"""
function trace_rays_frames(metric, objs::Vector{RayTraceGR.Object{T}}, canvases::Vector{RayTraceGR.Canvas{T}}; ctx = nothing) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, objs, ctx)
    if scene === nothing
        cpu_fallback("trace_rays", why)
        return [RayTraceGR.trace_rays(metric, objs, c) for c in canvases]
    end
    isempty(canvases) && return RayTraceGR.Canvas{T}[]
    ni, nj = size(canvases[1].pixels)
    all(c -> size(c.pixels) == (ni, nj), canvases) || error("trace_rays_frames: the canvases must have one size")
    opt = solver_of(T)
    outs = [similar(c.pixels) for c in canvases]
    pin = Ptr{Cvoid}[pointer(c.pixels) for c in canvases]
    pout = Ptr{Cvoid}[pointer(o) for o in outs]
    ctrs = Vector{RtgrCounters}(undef, length(canvases))
    GC.@preserve canvases outs begin
        if T === Float64
            check(ccall((:rtgr_trace_frames_pixels_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, UInt32, Ptr{Ptr{Cvoid}}, UInt64, UInt64, Ptr{Ptr{Cvoid}}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, length(canvases), pin, ni, nj, pout, ctrs))
        else
            check(ccall((:rtgr_trace_frames_pixels_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, UInt32, Ptr{Ptr{Cvoid}}, UInt64, UInt64, Ptr{Ptr{Cvoid}}, Ptr{RtgrCounters}),
                        handle(ctx), scene, opt, length(canvases), pin, ni, nj, pout, ctrs))
        end
    end
    [RayTraceGR.Canvas{T}(o) for o in outs]
end

"""
    trace_ray(metric, objs, cb, p::Pixel{T}; ctx = nothing) -> Pixel{T}

Legacy single-pixel shape (test/runtests.jl:76).  `cb` is ignored: the callback is always
`ContinuousCallback(min_distance(objs, ·), terminate!)` (src/RayTraceGR.jl:488-490).
"""
function trace_ray(metric, objs::Vector{RayTraceGR.Object{T}}, cb, p::RayTraceGR.Pixel{T}; ctx = nothing) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, objs, ctx)
    scene === nothing && error("trace_ray: ", why)
    opt = solver_of(T)
    pos, nrm = Ref(p.pos), Ref(p.normal)
    rgb = Ref(zeros(SVector{3,T})); se = Ref(zeros(SVector{8,T})); st = Ref{UInt8}(0)
    if T === Float64
        check(ccall((:rtgr_trace_one_f64, librtgr), Cint,
                    (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}),
                    handle(ctx), scene, opt, pos, nrm, rgb, se, st))
    else
        check(ccall((:rtgr_trace_one_f32, librtgr), Cint,
                    (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}),
                    handle(ctx), scene, opt, pos, nrm, rgb, se, st))
    end
    RayTraceGR.Pixel{T}(p.pos, p.normal, rgb[])
end

# ---- the reference's unit-test surface on the device (test/runtests.jl:12-61) --------------------------------------------
"""
    dmetric(metric, x::SVector{4,T}; ctx = nothing) -> (g::SMatrix{4,4,T}, dg::SArray{Tuple{4,4,4},T})
    christoffel(metric, x; ctx = nothing) -> Γ::SArray{Tuple{4,4,4},T}
    metric_at(metric, x; ctx = nothing) -> g

`RayTraceGR.dmetric` (:302-313) and `christoffel` (:321-331) evaluated by the device's dual-number path
(`rtgr_eval_metric_f64/_f32`) — what test/runtests.jl:12-61 exercises, `T = Float32` at :37.  The C arrays are row-major
`g[a][b]`, `dg[a][b][c] = ∂_c g_ab`, `Gam[a][b][c] = Γ^a_bc`; Julia's column-major view of the same bytes has the indices
reversed, hence the `permutedims`.
"""
function eval_metric(metric, x::SVector{4,T}, ctx) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, RayTraceGR.Object{T}[], ctx)
    scene === nothing && error("eval_metric: ", why)
    xs = collect(x)
    g = Array{T}(undef, 4, 4); dg = Array{T}(undef, 4, 4, 4); Γ = Array{T}(undef, 4, 4, 4)
    GC.@preserve xs g dg Γ begin
        if T === Float64
            check(ccall((:rtgr_eval_metric_f64, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{Float64}, UInt64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                        handle(ctx), scene, pointer(xs), 1, pointer(g), pointer(dg), pointer(Γ)))
        else
            check(ccall((:rtgr_eval_metric_f32, librtgr), Cint,
                        (Ctx, Ptr{RtgrScene}, Ptr{Float32}, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                        handle(ctx), scene, pointer(xs), 1, pointer(g), pointer(dg), pointer(Γ)))
        end
    end
    SMatrix{4,4,T}(permutedims(g)), SArray{Tuple{4,4,4},T}(permutedims(dg, (3, 2, 1))), SArray{Tuple{4,4,4},T}(permutedims(Γ, (3, 2, 1)))
end
metric_at(metric, x; ctx = nothing) = eval_metric(metric, x, ctx)[1]
dmetric(metric, x; ctx = nothing) = eval_metric(metric, x, ctx)[1:2]
christoffel(metric, x; ctx = nothing) = eval_metric(metric, x, ctx)[3]

"""
    geodesic(s::SVector{8,T}, metric; ctx = nothing, path = 2) -> ṡ::SVector{8,T}

`RayTraceGR.geodesic(r, metric, λ)` (:358-370) on the device (`rtgr_eval_geodesic_f64/_f32`): `path = 1` is the reference's own
formulation (duals → christoffel → contraction), `path = 2` exactly the function the production integrate loop calls.
"""
function geodesic(s::SVector{8,T}, metric; ctx = nothing, path::Integer = 2) where {T<:Union{Float64,Float32}}
    scene, why = scene_of(metric, RayTraceGR.Object{T}[], ctx)
    scene === nothing && error("geodesic: ", why)
    si = collect(s); so = similar(si)
    GC.@preserve si so begin
        if T === Float64
            check(ccall((:rtgr_eval_geodesic_f64, librtgr), Cint, (Ctx, Ptr{RtgrScene}, Ptr{Float64}, UInt64, Cint, Ptr{Float64}),
                        handle(ctx), scene, pointer(si), 1, path, pointer(so)))
        else
            check(ccall((:rtgr_eval_geodesic_f32, librtgr), Cint, (Ctx, Ptr{RtgrScene}, Ptr{Float32}, UInt64, Cint, Ptr{Float32}),
                        handle(ctx), scene, pointer(si), 1, path, pointer(so)))
        end
    end
    SVector{8,T}(so)
end

# ---- example1() / example2() (src/RayTraceGR.jl:540-612): drop-in twins that write the same PNGs ---------------------------
const outdir = "scenes"

"8-bit image the way `save(file, colorview(RGB, R', G', B'))` quantises it (N0f8: round(clamp(v, 0, 1) * 255), :566-575): `3 x ni x nj`"
quantize(planes) = UInt8[round(UInt8, clamp(planes[c][i, j], 0, 1) * 255) for c in 1:3, i in 1:size(planes[1], 1), j in 1:size(planes[1], 2)]

function save_scene(file, planes)
    # Images / ImageIO are the reference's own dependency for this step (:3-6, :575); loaded here so that `using RayTraceGRHIP`
    # alone does not pull them in
    Images = Base.require(Base.PkgId(Base.UUID("916415d5-f1e6-5110-898d-aaa5f9f070e0"), "Images"))
    T = eltype(planes[1])
    scene = Images.colorview(Images.RGB, T.(planes[1])', T.(planes[2])', T.(planes[3])')     # :566-569
    mkpath(dirname(file))
    rm(file, force = true)
    println("Output file is \"$file\"")
    Images.save(file, scene)                                                                 # :575
end

function example_scene(::Type{T}, which::Int) where {T}
    caelum = RayTraceGR.Sphere{T}(SVector{4,T}(0, 0, 0, 0), SVector{4,T}(1, 0, 0, 0), -10)              # :546, :584
    frustum = RayTraceGR.Plane{T}(-20)                                                                 # :547, :585
    centre = which == 1 ? SVector{4,T}(0, 0, 0, 0) : SVector{4,T}(0, 4, 0, 0)                          # :548, :586
    sphere = RayTraceGR.Sphere{T}(centre, SVector{4,T}(1, 0, 0, 0), T(1) / 2)
    objs = RayTraceGR.Object{T}[caelum, frustum, sphere]
    pos = which == 1 ? SVector{4,T}(0, 0, -2, 0) : SVector{4,T}(0, 4, -2, 0)                           # :554, :592
    objs, pos, SVector{4,T}(0, 1, 0, 0), SVector{4,T}(0, 0, 0, 1), SVector{4,T}(0, 0, 1, 0)            # widthx, widthy, normal
end

"""
    example1(; T = Float64, ni = 200, nj = 200, ctx = nothing, file = "scenes/sphere.png") -> Canvas{T}

`RayTraceGR.example1()` (:542-578) with `make_canvas` and `trace_rays` on the device: Minkowski, caelum + frustum + sphere,
the 200 x 200 screen — BASELINE config 1.  Writes the same PNG (39 855 of 40 000 pixels equal the committed `sphere.png`; the
rest is the silhouette ring, where the reference's own result depends on the last bit of the root-finder — SURVEY §4.2).
"""
function example1(; T::Type = Float64, ni::Int = 200, nj::Int = 200, ctx = nothing, file = joinpath(outdir, "sphere.png"))
    metric = RayTraceGR.minkowski
    objs, pos, widthx, widthy, normal = example_scene(T, 1)
    canvas = make_canvas(metric, pos, widthx, widthy, normal, ni, nj; ctx = ctx)
    canvas = trace_rays(metric, objs, canvas; ctx = ctx)
    file === nothing || save_scene(file, ntuple(c -> T[p.rgb[c] for p in canvas.pixels], 3))
    canvas
end

"""
    example2(; T = Float64, ni = 200, nj = 200, metric = kerr_schild, ctx = nothing, file = "scenes/sphere2.png") -> Canvas{T}

`RayTraceGR.example2()` (:580-612): Kerr–Schild, three objects.  Defaults reproduce the committed `sphere2.png` byte for byte.
BASELINE config 2: `example2(ni = 1024, nj = 1024, metric = KerrSchild(1.0, 0.8))`;
config 4: `example2(T = Float32, ni = 2048, nj = 2048, metric = KerrSchild(1.0, 0.8))`.
"""
function example2(; T::Type = Float64, ni::Int = 200, nj::Int = 200, metric = RayTraceGR.kerr_schild, ctx = nothing,
                  file = joinpath(outdir, "sphere2.png"))
    objs, pos, widthx, widthy, normal = example_scene(T, 2)
    canvas = make_canvas(metric, pos, widthx, widthy, normal, ni, nj; ctx = ctx)
    canvas = trace_rays(metric, objs, canvas; ctx = ctx)
    file === nothing || save_scene(file, ntuple(c -> T[p.rgb[c] for p in canvas.pixels], 3))
    canvas
end

"""
    example_disk(; a = 0.998, ni = 8192, nj = 8192, T = Float64, ctx = nothing, file = "scenes/disk.png") -> (R, G, B)

BASELINE config 5: near-extremal Kerr (`KerrSchild(1.0, a)`, textbook radius) with a thin accretion disk
`Disk(0.05, 2, 4)` in place of example2's small sphere (the camera, at cylindrical radius 4.5, stays outside the disk),
example2's camera, an 8192² screen — through `render` (camera on the device; an `Array{Pixel{Float64}}` of that size would be
5.9 GB each way).  The scene of `bench.py --variant ks_true0998_disk`.
"""
function example_disk(; a::Real = 0.998, ni::Int = 8192, nj::Int = 8192, T::Type = Float64, ctx = nothing,
                      file = joinpath(outdir, "disk.png"))
    objs, pos, widthx, widthy, normal = example_scene(T, 2)
    objs[3] = Disk{T}(T(0.05), T(2), T(4))
    planes = render(KerrSchild(1.0, a), objs, pos, widthx, widthy, normal, ni, nj; T = T, ctx = ctx)
    file === nothing || save_scene(file, planes)
    planes
end

end # module
