// tu_f64_ksref.hip — Float64 pipeline kernels of kerr_schild AS WRITTEN (RTGR_KS_REF; src/RayTraceGR.jl:274-294, r of :284):
// a = 0 (the reference's configuration, :276) and a != 0.
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f64_ksref(LaunchEnv& E, const TraceArgs<double>& A, bool spin, hipStream_t st) {
    return spin ? launch_trace<double, RTGR_KS_REF, true>(E, A, st) : launch_trace<double, RTGR_KS_REF, false>(E, A, st);
}
}  // namespace rtgr
