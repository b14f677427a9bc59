// tu_f64_generic.hip — Float64 pipeline with the GENERIC dual-number RHS (dmetric -> christoffel -> geodesic as the
// reference does it for any metric callable, src/RayTraceGR.jl:302-370): the built-in Kerr–Schild functions typed as
// metric functors, and the host side of run-time loaded user metrics (their kernels live in the user's code object).
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f64_generic(LaunchEnv& E, const TraceArgs<double>& A, hipStream_t st) {
    // a scene with a run-time unit — a metric of its own, or a built-in metric with user objects — launches the unit's kernels:
    // the RTGR_USER instantiation of the host-side sequence (it instantiates no integrate kernel itself)
    if (A.sc.metric == RTGR_USER || E.user) return launch_trace<double, RTGR_GENERIC_BASE + RTGR_USER, true>(E, A, st);
    if (A.sc.metric == RTGR_KS_REF) return launch_trace<double, RTGR_GENERIC_BASE + RTGR_KS_REF, true>(E, A, st);
    return launch_trace<double, RTGR_GENERIC_BASE + RTGR_KS_TRUE, true>(E, A, st);
}
}  // namespace rtgr
