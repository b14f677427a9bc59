# RayTraceGRHIP.jl — the reference-side binding a RayTraceGR.jl maintainer would add to route the hot path
# (`trace_rays`, src/RayTraceGR.jl:482-536) through librtgr_hip.so (include/rtgr.h).
#
# UNTESTED IN THIS REPOSITORY'S CI: the build image has no Julia.  It is a thin `ccall` layer; host code stays Julia,
# the metric/object/Pixel signatures of the reference are preserved, and anything that cannot cross the C ABI (an
# arbitrary metric callable) falls back to the reference's own CPU path.
module RayTraceGRHIP

using RayTraceGR
using StaticArrays

const librtgr = get(ENV, "RTGR_LIB", "librtgr_hip.so")
const RTGR_MAX_OBJECTS = 16

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

"""
    DeviceMetric(code_object; M = 1.0, a = 0.0)

A metric function of the user's own, given as a gfx950 code object built from `rtgr_user_unit.hip.in`
(INTEGRATION.md "A new metric") — the native stand-in for passing a new Julia function as `metric` (:302-309).
"""
struct DeviceMetric
    code_object::String
    M::Float64
    a::Float64
end
DeviceMetric(path; M = 1.0, a = 0.0) = DeviceMetric(path, M, a)
const resident_metric = Ref("")
function activate(m::DeviceMetric)
    resident_metric[] == m.code_object && return
    check(ccall((:rtgr_user_metric_load, librtgr), Cint, (Cstring,), m.code_object))
    resident_metric[] = m.code_object
end

metric_enum(m::DeviceMetric) = (activate(m); RTGR_USER)
metric_enum(m) = m === RayTraceGR.minkowski ? RTGR_MINKOWSKI :
                 m === RayTraceGR.kerr_schild ? RTGR_KS_REF : nothing   # as written: M = 1, a = 0 (:275-276)
metric_params(m::DeviceMetric) = (m.M, m.a)
metric_params(m) = (1.0, 0.0)

function check(rc)
    rc < 0 && error("librtgr_hip: ", unsafe_string(ccall((:rtgr_last_error, librtgr), Cstring, ())))
    rc
end

"""
    trace_rays(metric, objs, c::Canvas{Float64}) -> Canvas{Float64}

Drop-in for `RayTraceGR.trace_rays` (src/RayTraceGR.jl:483-484).  Passes `pointer(c.pixels)` — the reference's own
88-byte `Pixel{Float64}` AoS (:446-450) — across the ABI; returns a new canvas with `rgb` filled (:532).
"""
function trace_rays(metric, objs::Vector{RayTraceGR.Object{Float64}}, c::RayTraceGR.Canvas{Float64})
    me = metric_enum(metric)
    if me === nothing || length(objs) > RTGR_MAX_OBJECTS
        return RayTraceGR.trace_rays(metric, objs, c)        # arbitrary metric callable: reference CPU path
    end
    packed = ntuple(i -> i <= length(objs) ? pack(objs[i]) : NOOBJ, RTGR_MAX_OBJECTS)
    scene = Ref(RtgrScene(me, length(objs), metric_params(metric)..., packed))
    opt = Ref{RtgrSolver}()
    check(ccall((:rtgr_solver_defaults, librtgr), Cint, (Ptr{RtgrSolver}, Cint), opt, 0))
    ni, nj = size(c.pixels)
    out = similar(c.pixels)
    ctr = Ref{RtgrCounters}()
    GC.@preserve c out begin
        check(ccall((:rtgr_trace_pixels_f64, librtgr), Cint,
                    (Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, UInt64, UInt64, Ptr{Cvoid}, Ptr{RtgrCounters}),
                    scene, opt, pointer(c.pixels), ni, nj, pointer(out), ctr))
    end
    RayTraceGR.Canvas{Float64}(out)
end

"""
    trace_ray(metric, objs, cb, p::Pixel{Float64}) -> Pixel{Float64}

Legacy single-pixel shape (test/runtests.jl:76).  `cb` is ignored: the callback is always
`ContinuousCallback(min_distance(objs, ·), terminate!)` (src/RayTraceGR.jl:488-490).
"""
function trace_ray(metric, objs::Vector{RayTraceGR.Object{Float64}}, cb, p::RayTraceGR.Pixel{Float64})
    me = metric_enum(metric)
    me === nothing && error("only minkowski / kerr_schild cross the C ABI")
    packed = ntuple(i -> i <= length(objs) ? pack(objs[i]) : NOOBJ, RTGR_MAX_OBJECTS)
    scene = Ref(RtgrScene(me, length(objs), metric_params(metric)..., packed))
    opt = Ref{RtgrSolver}()
    check(ccall((:rtgr_solver_defaults, librtgr), Cint, (Ptr{RtgrSolver}, Cint), opt, 0))
    pos, nrm = Ref(p.pos), Ref(p.normal)
    rgb = Ref(zeros(SVector{3,Float64})); se = Ref(zeros(SVector{8,Float64})); st = Ref{UInt8}(0)
    check(ccall((:rtgr_trace_one_f64, librtgr), Cint,
                (Ptr{RtgrScene}, Ptr{RtgrSolver}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}),
                scene, opt, pos, nrm, rgb, se, st))
    RayTraceGR.Pixel{Float64}(p.pos, p.normal, rgb[])
end

end # module
