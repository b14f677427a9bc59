// tu_f64_kstrue.hip — Float64 pipeline kernels of the textbook Kerr–Schild metric (RTGR_KS_TRUE), a = 0 and a != 0.
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f64_kstrue(LaunchEnv& E, const TraceArgs<double>& A, bool spin, hipStream_t st) {
    return spin ? launch_trace<double, RTGR_KS_TRUE, true>(E, A, st) : launch_trace<double, RTGR_KS_TRUE, false>(E, A, st);
}
}  // namespace rtgr
