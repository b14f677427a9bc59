// tu_f64_generic.hip — Float64 pipeline with the GENERIC dual-number RHS (dmetric -> christoffel -> geodesic as the
// reference does it for any metric callable, src/RayTraceGR.jl:302-370): the built-in Kerr–Schild functions typed as
// metric functors, and the host side of run-time loaded user metrics (their kernels live in the user's code object).
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f64_generic(LaunchEnv& E, const TraceArgs<double>& A, hipStream_t st) {
    if (A.sc.metric == RTGR_USER) return launch_trace<double, RTGR_GENERIC_BASE + RTGR_USER, true>(E, A, st);
    if (A.sc.metric == RTGR_KS_REF) return launch_trace<double, RTGR_GENERIC_BASE + RTGR_KS_REF, true>(E, A, st);
    return launch_trace<double, RTGR_GENERIC_BASE + RTGR_KS_TRUE, true>(E, A, st);
}
}  // namespace rtgr
