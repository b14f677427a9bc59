// Measures the accuracy of the v_rcp_f64 / v_rsq_f64 hardware seeds and of the Newton-refined forms used by
// rtgr_physics.hpp (frcp / frsq), against correctly rounded host results.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double e = __builtin_fma(-v, r0, 1.0);
    double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-v, r1, 1.0);
    double r2 = __builtin_fma(r1, e, r1);
    double s0 = __builtin_amdgcn_rsq(v);
    double f = __builtin_fma(-v * s0, s0, 1.0);
    double s1 = __builtin_fma(0.5 * s0, f, s0);
    f = __builtin_fma(-v * s1, s1, 1.0);
    double s2 = __builtin_fma(0.5 * s1, f, s1);
    // one third-order step from the seed: s(1 + e/2 + 3e²/8)
    double g = __builtin_fma(-v * s0, s0, 1.0);
    double h3 = __builtin_fma(s0 * g, __builtin_fma(0.375, g, 0.5), s0);
    o[7 * i + 0] = r0; o[7 * i + 1] = r1; o[7 * i + 2] = r2;
    o[7 * i + 3] = s0; o[7 * i + 4] = s1; o[7 * i + 5] = s2; o[7 * i + 6] = h3;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(7 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-10, 10);
    for (auto& v : x) v = std::exp2(u(g));
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 7 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 7 * n * 8, hipMemcpyDeviceToHost);
    double m[7] = {0};
    for (int i = 0; i < n; i++) {
        long double rc = 1.0L / x[i], rs = 1.0L / sqrtl((long double)x[i]);
        for (int j = 0; j < 7; j++) {
            long double ref = j < 3 ? rc : rs;
            double err = (double)fabsl((o[7 * i + j] - ref) / ref);
            if (err > m[j]) m[j] = err;
        }
    }
    printf("max rel err: rcp seed %.3g (2^%.1f)  +1NR %.3g  +2NR %.3g | rsq seed %.3g (2^%.1f)  +1NR %.3g  +2NR %.3g  1x3rd-order %.3g\n",
           m[0], log2(m[0]), m[1], m[2], m[3], log2(m[3]), m[4], m[5], m[6]);
    return 0;
}
