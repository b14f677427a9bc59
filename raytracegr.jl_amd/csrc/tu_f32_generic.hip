// tu_f32_generic.hip — Float32 pipeline with the generic dual-number RHS: what the reference's own test pushes through
// kerr_schild + dmetric (T = Float32, test/runtests.jl:37-60), end to end.
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f32_generic(LaunchEnv& E, const TraceArgs<float>& A, hipStream_t st) {
    // a scene with a run-time unit — a metric of its own, or a built-in metric with user objects — launches the unit's kernels:
    // the RTGR_USER instantiation of the host-side sequence (it instantiates no integrate kernel itself)
    if (A.sc.metric == RTGR_USER || E.user) return launch_trace<float, RTGR_GENERIC_BASE + RTGR_USER, true>(E, A, st);
    if (A.sc.metric == RTGR_KS_REF) return launch_trace<float, RTGR_GENERIC_BASE + RTGR_KS_REF, true>(E, A, st);
    return launch_trace<float, RTGR_GENERIC_BASE + RTGR_KS_TRUE, true>(E, A, st);
}
}  // namespace rtgr
