# RayTraceGRHIP.jl — the reference-side binding a RayTraceGR.jl maintainer would add to route the hot path
# (`trace_rays`, src/RayTraceGR.jl:482-536) through librtgr_hip.so (include/rtgr.h, ABI version 2).
#
# NOT EXECUTED IN THIS REPOSITORY: the build image has no Julia.  What stands in for running it:
#   * tests/c/abi_layout.c — a compiled C caller that passes the same bytes this file would (structs by pointer, an
#     88-byte Pixel array) and whose _Static_asserts pin the table below; tests/test_abi.py runs it on the CPU (layout,
#     symbols) and on the GPU (example2() == sphere2.png through rtgr_trace_pixels_f64 + rtgr_trace_one_f64);
#   * raytracegr.jl_amd/api.py — the same calls through Python ctypes, which every GPU test uses;
#   * tests/test_julia_stub.py — every `ccall` below against the prototypes of include/rtgr.h (symbol declared and exported,
#     as many argument types and arguments as C parameters, pointer / Cint / UInt64 kinds, return type), the structs' field
#     order against the header's, and the block structure of this file.
#
# fieldoffset table (bytes; Julia lays isbits structs out by the C rules, so `fieldoffset(T, i)` must print exactly this —
# a maintainer can check with `[(fieldname(T,i), fieldoffset(T,i)) for i in 1:fieldcount(T)]`):
#
#   RtgrObject       80   kind 0, reserved 4, p 8
#   RtgrScene     1312   metric 0, nobj 4, M 8, a 16, user_metric 24, obj 32
#   RtgrSolver      72   reltol 0, abstol 8, lambda0 16, lambda1 24, hit_threshold 32, miss_rgb 40, max_steps 64, interp_points 68
#   RtgrCounters    64   rays 0, accepted 8, rejected 16, rhs_evals 24, events 32, events_interior 40, not_finished 48, reserved 56
#   Pixel{Float64}  88   pos 0, normal 32, rgb 64          (the reference's own type, src/RayTraceGR.jl:446-450)
#   Pixel{Float32}  44   pos 0, normal 16, rgb 32
#
# It is a thin `ccall` layer; host code stays Julia, the metric/object/Pixel signatures of the reference are preserved,
# and anything that cannot cross the C ABI (an arbitrary metric callable) falls back to the reference's own CPU path.
module RayTraceGRHIP

using RayTraceGR
using StaticArrays

const librtgr = get(ENV, "RTGR_LIB", "librtgr_hip.so")
const RTGR_MAX_OBJECTS = 16
const Ctx = Ptr{Cvoid}          # rtgr_context*; C_NULL = the process's default context

# ---- PODs of include/rtgr.h ------------------------------------------------------------------------------------------
struct RtgrObject
    kind::UInt32
    reserved::UInt32
    p::NTuple{9,Float64}
end
struct RtgrScene
    metric::UInt32
    nobj::UInt32
    M::Float64
    a::Float64
    user_metric::UInt64
    obj::NTuple{RTGR_MAX_OBJECTS,RtgrObject}
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
struct RtgrCounters
    rays::UInt64; accepted::UInt64; rejected::UInt64; rhs_evals::UInt64
    events::UInt64; events_interior::UInt64; not_finished::UInt64; reserved::UInt64
end

const RTGR_MINKOWSKI, RTGR_KS_REF, RTGR_KS_TRUE, RTGR_USER = UInt32(0), UInt32(1), UInt32(2), UInt32(3)
const RTGR_PLANE, RTGR_SPHERE = UInt32(1), UInt32(2)

pack(pl::RayTraceGR.Plane{Float64}) = RtgrObject(RTGR_PLANE, 0, (pl.time, 0, 0, 0, 0, 0, 0, 0, 0))
pack(s::RayTraceGR.Sphere{Float64}) = RtgrObject(RTGR_SPHERE, 0, (s.pos..., s.vel..., s.radius))
const NOOBJ = RtgrObject(0, 0, ntuple(_ -> 0.0, 9))

function check(rc)
    rc < 0 && error("librtgr_hip: ", unsafe_string(ccall((:rtgr_last_error, librtgr), Cstring, ())))
    rc
end

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

"""
    DeviceMetric(code_object; M = 1.0, a = 0.0)

A metric function of the user's own — the native stand-in for passing a new Julia function as `metric` (:302-309) —
given as C++ source text (`DeviceMetric(source = "...")`: compiled in-process with hiprtc by `rtgr_user_metric_compile`) or
as a gfx950 code object built from `rtgr_user_unit.hip.in` (INTEGRATION.md "A new metric").
Several may be resident at once; a scene names its own by id.
"""
struct DeviceMetric
    code_object::String     # path of a code object built with hipcc --genco, or "" when `source` is given
    source::String          # C++ source of rtgr_user_metric<S>: compiled in-process by the library (hiprtc), one ccall
    stationary::Bool
    M::Float64
    a::Float64
end
DeviceMetric(path::AbstractString; M = 1.0, a = 0.0) = DeviceMetric(path, "", false, M, a)
DeviceMetric(; source::AbstractString, stationary = false, M = 1.0, a = 0.0) = DeviceMetric("", source, stationary, M, a)
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
metric_desc(m, ctx) = m === RayTraceGR.minkowski ? (RTGR_MINKOWSKI, 1.0, 0.0, UInt64(0)) :
                      m === RayTraceGR.kerr_schild ? (RTGR_KS_REF, 1.0, 0.0, UInt64(0)) :   # as written: M = 1, a = 0 (:275-276)
                      nothing

function scene_of(metric, objs, ctx)
    d = metric_desc(metric, ctx)
    (d === nothing || length(objs) > RTGR_MAX_OBJECTS) && return nothing
    packed = ntuple(i -> i <= length(objs) ? pack(objs[i]) : NOOBJ, RTGR_MAX_OBJECTS)
    Ref(RtgrScene(d[1], length(objs), d[2], d[3], d[4], packed))
end

# the objects' own parameters stay Float64 across the ABI (rtgr_object.p); T selects the arithmetic of the path
pack(pl::RayTraceGR.Plane{Float32}) = RtgrObject(RTGR_PLANE, 0, (Float64(pl.time), 0, 0, 0, 0, 0, 0, 0, 0))
pack(s::RayTraceGR.Sphere{Float32}) = RtgrObject(RTGR_SPHERE, 0, (Float64.(s.pos)..., Float64.(s.vel)..., Float64(s.radius)))

"""
    trace_rays(metric, objs, c::Canvas{T}; ctx = nothing) -> Canvas{T},   T = Float64 | Float32

Drop-in for `RayTraceGR.trace_rays` (src/RayTraceGR.jl:483-484).  Passes `pointer(c.pixels)` — the reference's own
`Pixel{T}` AoS (:446-450; 88 bytes for Float64, 44 for Float32) — across the ABI; returns a new canvas with `rgb`
filled (:532).  The tolerance is `eps(T)^(3/4)` as in the reference (:485).  `ctx = Context(0:7)`: all eight GPUs of a
node work on the canvas (rows dealt cyclically); the result does not depend on the number of devices, bit for bit.
"""
function trace_rays(metric, objs::Vector{RayTraceGR.Object{T}}, c::RayTraceGR.Canvas{T}; ctx = nothing) where {T<:Union{Float64,Float32}}
    scene = scene_of(metric, objs, ctx)
    scene === nothing && return RayTraceGR.trace_rays(metric, objs, c)        # arbitrary metric callable: reference CPU path
    opt = Ref{RtgrSolver}()
    check(ccall((:rtgr_solver_defaults, librtgr), Cint, (Ptr{RtgrSolver}, Cint), opt, T === Float32 ? 1 : 0))
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

"""
    trace_ray(metric, objs, cb, p::Pixel{Float64}; ctx = nothing) -> Pixel{Float64}

Legacy single-pixel shape (test/runtests.jl:76).  `cb` is ignored: the callback is always
`ContinuousCallback(min_distance(objs, ·), terminate!)` (src/RayTraceGR.jl:488-490).
"""
function trace_ray(metric, objs::Vector{RayTraceGR.Object{Float64}}, cb, p::RayTraceGR.Pixel{Float64}; ctx = nothing)
    scene = scene_of(metric, objs, ctx)
    scene === nothing && error("only minkowski / kerr_schild / DeviceMetric cross the C ABI")
    opt = Ref{RtgrSolver}()
    check(ccall((:rtgr_solver_defaults, librtgr), Cint, (Ptr{RtgrSolver}, Cint), opt, 0))
    pos, nrm = Ref(p.pos), Ref(p.normal)
    rgb = Ref(zeros(SVector{3,Float64})); se = Ref(zeros(SVector{8,Float64})); st = Ref{UInt8}(0)
    check(ccall((:rtgr_trace_one_f64, librtgr), Cint,
                (Ctx, Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}),
                handle(ctx), scene, opt, pos, nrm, rgb, se, st))
    RayTraceGR.Pixel{Float64}(p.pos, p.normal, rgb[])
end

end # module
