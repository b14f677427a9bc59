// rtgr_misc.hip — the small kernels around the pipeline and their launchers:
//   canvas_kernel<R>              make_canvas (src/RayTraceGR.jl:457-478)
//   eval_metric_kernel / eval_geodesic_kernel / eval_fastmath_kernel   parity hooks (the reference's unit-test surface,
//                                 test/runtests.jl:12-61; the hot loop's own RHS and reciprocal helpers)
//   quantize_kernel               N0f8 rounding + transposed image layout of save() (:566-575)
//   redshift_kernel               frequency ratio observed/emitted from Sphere.vel (no reference counterpart)
//   place_rows                    multi-device gather: a rank's cyclic rows back into the full frame on device 0
#include "rtgr_host.hpp"
#include "rtgr_integrator.hpp"

namespace rtgr {

static inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

template <class R>
__global__ __launch_bounds__(256) void canvas_kernel(DevScene<R> sc, DevCamera<R> cam, uint64_t ni, uint64_t nj,
                                                     uint64_t j0, uint64_t jstride, uint64_t first, uint64_t count, R* state0) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= count) return;
    const uint64_t idx = first + w;  // linear index inside the slab: i + k * ni, image row j = j0 + k * jstride
    R s[8];
    make_pixel<R>(sc, cam, ni, nj, idx % ni, j0 + (idx / ni) * jstride, s);
#pragma unroll
    for (int c = 0; c < 8; c++) state0[w * 8 + c] = s[c];
}

template <class R>
__global__ __launch_bounds__(256) void eval_metric_kernel(DevScene<R> sc, const R* x, uint64_t n, R* g, R* dg, R* Gam) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    R xx[4] = {x[4 * p], x[4 * p + 1], x[4 * p + 2], x[4 * p + 3]};
    R gg[4][4], dd[4][4][4];
    dmetric_dev<R>(sc.metric, sc.M, sc.a, xx, gg, dd);
    if (g) for (int q = 0; q < 16; q++) g[16 * p + q] = (&gg[0][0])[q];
    if (dg) for (int q = 0; q < 64; q++) dg[64 * p + q] = (&dd[0][0][0])[q];
    if (Gam) {
        R GG[4][4][4];
        christoffel_dev<R>(gg, dd, GG);
        for (int q = 0; q < 64; q++) Gam[64 * p + q] = (&GG[0][0][0])[q];
    }
}

// path 0: the Kerr–Schild-form contraction with IEEE division / sqrt (ks_field + ksform_accel: what the tile kernel runs)
// path 1: the generic dual-number RHS (what RTGR_METRIC_GENERIC and user metrics run)
// path 2: EXACTLY the function the production integrate loop calls — accel<R, METRIC, SPIN, FAST = true>
//         (accel_radial / accel_spin_true / accel_spin_ref with the 4/6-instruction frcp / frsq, the KS_TRUE null-congruence shortcuts)
template <class R, int METRIC, bool SPIN>
RTGR_DEV void rhs_paths(const R* si, R M, R a, int path, R* so) {
    if (path == 2) {
        so[0] = si[4]; so[1] = si[5]; so[2] = si[6]; so[3] = si[7];
        accel<R, METRIC, SPIN, true>(si + 1, si + 4, metric_consts<R>(M, a), so + 4);
    } else {
        rhs<R, METRIC, SPIN>(si, M, a, so);
    }
}

template <class R>
__global__ __launch_bounds__(256) void eval_geodesic_kernel(DevScene<R> sc, const R* s, uint64_t n, int path, R* ds) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    R si[8], so[8];
    for (int c = 0; c < 8; c++) si[c] = s[8 * p + c];
    if (path == 1) {
        generic_rhs<R>(sc.metric, sc.M, sc.a, si, so);
    } else {
        const bool spin = sc.a != R(0);
        if (sc.metric == RTGR_MINKOWSKI) rhs_paths<R, RTGR_MINKOWSKI, false>(si, sc.M, sc.a, path, so);
        else if (sc.metric == RTGR_KS_REF) {
            if (spin) rhs_paths<R, RTGR_KS_REF, true>(si, sc.M, sc.a, path, so);
            else rhs_paths<R, RTGR_KS_REF, false>(si, sc.M, sc.a, path, so);
        } else {
            if (spin) rhs_paths<R, RTGR_KS_TRUE, true>(si, sc.M, sc.a, path, so);
            else rhs_paths<R, RTGR_KS_TRUE, false>(si, sc.M, sc.a, path, so);
        }
    }
    for (int c = 0; c < 8; c++) ds[8 * p + c] = so[c];
}

template <class R>
__global__ __launch_bounds__(256) void eval_objects_kernel(DevScene<R> sc, DevSolver<R> opt, const R* x, uint64_t n, R* d, R* dmin, uint8_t* hit, R* rgb) {
    eval_objects_body<R>(sc, opt, x, n, d, dmin, hit, rgb);
}

// the hot loop's reciprocal / reciprocal-square-root helpers (rtgr_physics.hpp: frcp, frsq), exposed so that their
// accuracy claim (<= 1.5e-16 relative) is a test, not a comment
__global__ __launch_bounds__(256) void eval_fastmath_kernel(const double* x, uint64_t n, double* rcp, double* rsq) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (rcp) rcp[p] = frcp<double>(x[p]);
    if (rsq) rsq[p] = frsq<double>(x[p]);
}

// N0f8 quantisation (round(255 x), FixedPointNumbers) + transposed layout image[j][i][c]
__global__ __launch_bounds__(256) void quantize_kernel(const double* rgb, uint64_t ni, uint64_t nj, uint8_t* img) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = ni * nj;
    if (idx >= n) return;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double v = rgb[c * n + idx];
        v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        img[idx * 3 + c] = (uint8_t)__builtin_rint(v * 255.0);  // idx = i + j*ni  ==  row j, column i
    }
}

// part: `planes` planes of ni*nrows elements, local row k = image row rank + k*nranks;  full: planes of ni*nj
template <class T>
__global__ __launch_bounds__(256) void place_rows_kernel(const T* part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks,
                                                         uint64_t planes, uint64_t elem, T* full) {
    const uint64_t nrows = (nj - rank + nranks - 1) / nranks;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_plane = ni * nrows * elem;
    if (t >= per_plane * planes) return;
    const uint64_t pl = t / per_plane, r = t - pl * per_plane;
    const uint64_t row_elems = ni * elem;
    const uint64_t k = r / row_elems, i = r - k * row_elems;
    full[pl * ni * nj * elem + (rank + k * nranks) * row_elems + i] = part[t];
}

// ---- redshift: redshift_body (rtgr_integrator.hpp) ----------------------------------------------------------------------
template <class R>
__global__ __launch_bounds__(256) void redshift_kernel(DevScene<R> sc, DevCamera<R> cam, const R* state0, uint64_t ni, uint64_t nj,
                                                       uint64_t j0, uint64_t jstride, uint64_t n, uint64_t out_offset,
                                                       const R* state_end, const uint8_t* hit, const uint32_t* hit32, R* red) {
    redshift_body<R>(sc, cam, state0, ni, nj, j0, jstride, n, out_offset, state_end, hit, hit32, red);
}

#define CHECK_LAUNCH()                                     \
    do {                                                   \
        hipError_t e_ = hipGetLastError();                 \
        if (e_ != hipSuccess) return fail(RTGR_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e_)); \
    } while (0)

int misc_canvas_f64(const DevScene<double>& sc, const DevCamera<double>& cam, uint64_t ni, uint64_t nj, uint64_t j0,
                    uint64_t n, double* d_state0, hipStream_t st) {
    hipLaunchKernelGGL(canvas_kernel<double>, dim3(nblk(n)), dim3(256), 0, st, sc, cam, ni, nj, j0, (uint64_t)1, (uint64_t)0, n, d_state0);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_canvas_f32(const DevScene<float>& sc, const DevCamera<float>& cam, uint64_t ni, uint64_t nj, uint64_t j0,
                    uint64_t n, float* d_state0, hipStream_t st) {
    hipLaunchKernelGGL(canvas_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, sc, cam, ni, nj, j0, (uint64_t)1, (uint64_t)0, n, d_state0);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_metric_f64(const DevScene<double>& sc, const double* d_x, uint64_t n, double* g, double* dg, double* Gam, hipStream_t st) {
    hipLaunchKernelGGL(eval_metric_kernel<double>, dim3(nblk(n)), dim3(256), 0, st, sc, d_x, n, g, dg, Gam);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_metric_f32(const DevScene<float>& sc, const float* d_x, uint64_t n, float* g, float* dg, float* Gam, hipStream_t st) {
    hipLaunchKernelGGL(eval_metric_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, sc, d_x, n, g, dg, Gam);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_objects_f64(const DevScene<double>& sc, const DevSolver<double>& opt, const double* d_x, uint64_t n, double* d, double* dmin, uint8_t* hit,
                          double* rgb, hipStream_t st) {
    hipLaunchKernelGGL(eval_objects_kernel<double>, dim3(nblk(n)), dim3(256), 0, st, sc, opt, d_x, n, d, dmin, hit, rgb);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_objects_f32(const DevScene<float>& sc, const DevSolver<float>& opt, const float* d_x, uint64_t n, float* d, float* dmin, uint8_t* hit,
                          float* rgb, hipStream_t st) {
    hipLaunchKernelGGL(eval_objects_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, sc, opt, d_x, n, d, dmin, hit, rgb);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_geodesic_f64(const DevScene<double>& sc, const double* d_s, uint64_t n, int path, double* d_ds, hipStream_t st) {
    hipLaunchKernelGGL(eval_geodesic_kernel<double>, dim3(nblk(n)), dim3(256), 0, st, sc, d_s, n, path, d_ds);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_geodesic_f32(const DevScene<float>& sc, const float* d_s, uint64_t n, int path, float* d_ds, hipStream_t st) {
    hipLaunchKernelGGL(eval_geodesic_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, sc, d_s, n, path, d_ds);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_eval_fastmath_f64(const double* d_x, uint64_t n, double* d_rcp, double* d_rsq, hipStream_t st) {
    hipLaunchKernelGGL(eval_fastmath_kernel, dim3(nblk(n)), dim3(256), 0, st, d_x, n, d_rcp, d_rsq);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_quantize(const double* d_rgb, uint64_t ni, uint64_t nj, uint8_t* d_img, hipStream_t st) {
    hipLaunchKernelGGL(quantize_kernel, dim3(nblk(ni * nj)), dim3(256), 0, st, d_rgb, ni, nj, d_img);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_place_rows_f64(const double* d_part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks, uint64_t planes,
                        double* d_full, hipStream_t st) {
    const uint64_t nrows = (nj - rank + nranks - 1) / nranks;
    hipLaunchKernelGGL(place_rows_kernel<double>, dim3(nblk(ni * nrows * planes)), dim3(256), 0, st, d_part, ni, nj, rank, nranks,
                       planes, (uint64_t)1, d_full);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_place_rows_f32(const float* d_part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks, uint64_t planes,
                        float* d_full, hipStream_t st) {
    const uint64_t nrows = (nj - rank + nranks - 1) / nranks;
    hipLaunchKernelGGL(place_rows_kernel<float>, dim3(nblk(ni * nrows * planes)), dim3(256), 0, st, d_part, ni, nj, rank, nranks,
                       planes, (uint64_t)1, d_full);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_place_rows_u8(const uint8_t* d_part, uint64_t ni, uint64_t nj, uint64_t rank, uint64_t nranks, uint64_t elem,
                       uint8_t* d_full, hipStream_t st) {
    const uint64_t nrows = (nj - rank + nranks - 1) / nranks;
    hipLaunchKernelGGL(place_rows_kernel<uint8_t>, dim3(nblk(ni * nrows * elem)), dim3(256), 0, st, d_part, ni, nj, rank, nranks,
                       (uint64_t)1, elem, d_full);
    CHECK_LAUNCH();
    return RTGR_OK;
}

}  // namespace rtgr

namespace rtgr {
int misc_redshift_f64(const DevScene<double>& sc, const DevCamera<double>& cam, const double* d_state0, uint64_t ni, uint64_t nj,
                      uint64_t j0, uint64_t jstride, uint64_t n, uint64_t out_offset, const double* d_state_end,
                      const uint8_t* d_hit, const uint32_t* d_hit32, double* d_red, hipStream_t st) {
    hipLaunchKernelGGL(redshift_kernel<double>, dim3(nblk(n)), dim3(256), 0, st, sc, cam, d_state0, ni, nj, j0, jstride, n, out_offset,
                       d_state_end, d_hit, d_hit32, d_red);
    CHECK_LAUNCH();
    return RTGR_OK;
}
int misc_redshift_f32(const DevScene<float>& sc, const DevCamera<float>& cam, const float* d_state0, uint64_t ni, uint64_t nj,
                      uint64_t j0, uint64_t jstride, uint64_t n, uint64_t out_offset, const float* d_state_end,
                      const uint8_t* d_hit, const uint32_t* d_hit32, float* d_red, hipStream_t st) {
    hipLaunchKernelGGL(redshift_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, sc, cam, d_state0, ni, nj, j0, jstride, n, out_offset,
                       d_state_end, d_hit, d_hit32, d_red);
    CHECK_LAUNCH();
    return RTGR_OK;
}
// The load-time probe's scrubber (rtgr_units.hip: probe_trace).  The compiler fault the probe exists for makes a kernel read registers it
// never wrote for some lanes — whatever the wave that had the SIMD before left there.  When that was an identical run of the same frame,
// the stale values are the RIGHT ones and the fault hides (round 6: the probe passed or refused a faulty unit depending on which tests
// had run before it).  So every probe run starts from registers that hold a pattern of ITS OWN: one wave per launch slot that owns a
// SIMD's whole register file (256 VGPRs + 256 AGPRs x 64 lanes = 128 KB) and writes the pattern into all of it; eight waves per SIMD
// of the device, so that every SIMD is visited.
__global__ __launch_bounds__(64) void poison_registers_kernel(unsigned pattern) {
    // (… and 64 KB of the CU's LDS with it: a skipped LDS store finds the previous run's value just the same)
    __shared__ unsigned lds[16384];
    for (unsigned k = threadIdx.x; k < 16384u; k += 64u) ((volatile unsigned*)lds)[k] = pattern;
    asm volatile(
        ".set rtgr_poison_i, 0\n"
        ".rept 256\n"
        "v_mov_b32 v[rtgr_poison_i], %0\n"
        "s_nop 1\n"
        "v_accvgpr_write_b32 a[rtgr_poison_i], v[rtgr_poison_i]\n"
        ".set rtgr_poison_i, rtgr_poison_i + 1\n"
        ".endr\n"
        "s_nop 4\n"
        :
        : "s"(pattern)
        : "v255", "a255", "memory");
}
int misc_poison_registers(int n_cu, unsigned pattern, hipStream_t st) {
    hipLaunchKernelGGL(poison_registers_kernel, dim3((unsigned)(n_cu > 0 ? n_cu : 256) * 4u * 8u), dim3(64), 0, st, pattern);
    CHECK_LAUNCH();
    return RTGR_OK;
}

}  // namespace rtgr
