// tu_f64_mink.hip — Float64 pipeline kernels of minkowski (src/RayTraceGR.jl:262-264): Γ ≡ 0.
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f64_mink(LaunchEnv& E, const TraceArgs<double>& A, hipStream_t st) {
    return launch_trace<double, RTGR_MINKOWSKI, false>(E, A, st);
}
}  // namespace rtgr
