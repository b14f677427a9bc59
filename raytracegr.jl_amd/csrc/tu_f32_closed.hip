// tu_f32_closed.hip — Float32 pipeline kernels (BASELINE config 4) of the three built-in metrics, closed contraction.
#include "rtgr_pipeline.hpp"
namespace rtgr {
int launch_f32_closed(LaunchEnv& E, const TraceArgs<float>& A, bool spin, hipStream_t st) {
    switch (A.sc.metric) {
        case RTGR_MINKOWSKI: return launch_trace<float, RTGR_MINKOWSKI, false>(E, A, st);
        case RTGR_KS_REF:
            return spin ? launch_trace<float, RTGR_KS_REF, true>(E, A, st) : launch_trace<float, RTGR_KS_REF, false>(E, A, st);
        default:
            return spin ? launch_trace<float, RTGR_KS_TRUE, true>(E, A, st) : launch_trace<float, RTGR_KS_TRUE, false>(E, A, st);
    }
}
}  // namespace rtgr
