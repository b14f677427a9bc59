// exec_flip_repro.hip — a stand-alone kernel on which ROCm 7.2's compiler (AMD clang 22.0.0git, roc-7.2.0, gfx950) places register
// copies AHEAD of the instruction that switches EXEC to the `else` lanes of a divergent if / else (the rule and the
// repair: raytracegr.jl_amd/isa_exec.py).  Nothing of this repository is needed to see it:
//
//     hipcc -w --cuda-device-only --offload-arch=gfx950 -O3 -DWAVES=1 -DEXTRA=40 -S tools/micro/exec_flip_repro.hip -o repro.s
//     grep -n -B12 "s_andn2_saveexec_b64" repro.s | less        # look for v_accvgpr_write_b32 / scratch_store between a label that an
//                                                               # s_cbranch_execz targets and the s_andn2_saveexec_b64 that follows it
// e.g. (this compiler, -DWAVES=1 -DEXTRA=40):    .LBB0_124:                              ; <- target of `s_cbranch_execz .LBB0_124`
//                                                    v_accvgpr_write_b32 a10, v174       ; runs for the THEN lanes only (usually none)
//                                                    ...
//                                                    s_andn2_saveexec_b64 s[0:1], s[2:3] ; EXEC = the else lanes, only now
// and -DWAVES=2 -DEXTRA=40 shows the same with scratch_store_dwordx2 / scratch_load_dwordx2 / v_mov_b64 in that place.
// tools/micro/exec_flip_repro.sh builds the kernel as compiled and once more with the block rewritten (isa_exec.repair), runs both on
// the same inputs (this file's main) and prints where their results differ — expected (repaired) against actual (as compiled).
//
// What it takes (found by construction, not by reduction): a body with many FLOW blocks that have constants hoisted into them — the
// inlined OCML sin / cos / acos / atan2 / pow / exp / log / cbrt / atan on forward duals — inside a loop that keeps enough values
// live across them that the allocator splits live ranges into AGPRs (one wave per SIMD) or spills (two).  EXTRA adds live values.
#include <hip/hip_runtime.h>
#ifndef WAVES
#define WAVES 1
#endif
#ifndef EXTRA
#define EXTRA 0
#endif
struct D3 { double v, e[3]; };   // forward dual, three partials
#define DEV static __device__ __forceinline__
DEV D3 mk(double v) { return D3{v, {0, 0, 0}}; }
DEV D3 operator+(D3 a, D3 b) { return D3{a.v + b.v, {a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]}}; }
DEV D3 operator*(D3 a, D3 b) { return D3{a.v * b.v, {a.e[0] * b.v + a.v * b.e[0], a.e[1] * b.v + a.v * b.e[1], a.e[2] * b.v + a.v * b.e[2]}}; }
DEV D3 operator*(double s, D3 a) { return D3{s * a.v, {s * a.e[0], s * a.e[1], s * a.e[2]}}; }
DEV D3 chain(D3 x, double f, double df) { return D3{f, {df * x.e[0], df * x.e[1], df * x.e[2]}}; }
DEV D3 dsqrt(D3 x) { const double r = sqrt(x.v); return chain(x, r, 0.5 / r); }
DEV D3 drcp(D3 x) { const double r = 1.0 / x.v; return chain(x, r, -r * r); }
DEV D3 dsin(D3 x) { return chain(x, sin(x.v), cos(x.v)); }
DEV D3 dcos(D3 x) { return chain(x, cos(x.v), -sin(x.v)); }
DEV D3 dacos(D3 x) { return chain(x, acos(x.v), -1.0 / sqrt(1.0 - x.v * x.v)); }
DEV D3 datan2(D3 y, D3 x) { const double q = 1.0 / (x.v * x.v + y.v * y.v); D3 r{atan2(y.v, x.v), {0, 0, 0}};
    for (int i = 0; i < 3; i++) r.e[i] = (x.v * y.e[i] - y.v * x.e[i]) * q; return r; }
DEV D3 dpow(D3 x, double p) { const double f = pow(x.v, p); return chain(x, f, p * f / x.v); }
DEV D3 dexp(D3 x) { const double f = exp(x.v); return chain(x, f, f); }
DEV D3 dlog(D3 x) { return chain(x, log(x.v), 1.0 / x.v); }
DEV D3 dcbrt(D3 x) { const double c = cbrt(x.v); return chain(x, c, 1.0 / (3.0 * c * c)); }
DEV D3 datan(D3 x) { return chain(x, atan(x.v), 1.0 / (1.0 + x.v * x.v)); }
// "acceleration" at a point: a few metric-like entries through every elementary function, contracted with u
DEV void accel(const double x[3], const double u[4], double out[4]) {
    D3 X{x[0], {1, 0, 0}}, Y{x[1], {0, 1, 0}}, Z{x[2], {0, 0, 1}};
    const D3 rho = dsqrt(X * X + Y * Y + Z * Z), irho = drcp(rho), cth = Z * irho;
    const D3 th = dacos(cth), ph = datan2(Y, X);
    const D3 g00 = mk(-1.0) + (-0.1) * dpow(rho, -1.5), g11 = mk(1.0) + 0.02 * dexp((-1.0) * rho), g22 = mk(1.0) + 0.05 * dcbrt(mk(1.0) + rho) * irho;
    const D3 g33 = mk(1.0) + 0.03 * datan(rho) * dlog(mk(2.0) + rho) * irho, g12 = 0.01 * dsin(ph) * dcos(th), g03 = 0.02 * cth * irho;
    for (int j = 0; j < 3; j++)
        out[1 + j] = g00.e[j] * u[0] * u[0] + g11.e[j] * u[1] * u[1] + g22.e[j] * u[2] * u[2] + g33.e[j] * u[3] * u[3] + 2.0 * g12.e[j] * u[1] * u[2] + 2.0 * g03.e[j] * u[0] * u[3];
    out[0] = g00.v * u[0] + g03.v * u[3] + g11.v + g22.v + g33.v + g12.v;
}
extern "C" __global__ __launch_bounds__(64, WAVES) void repro(const double* in, double* out, int n, int steps) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double x[3], u[4], k[7][4];
    for (int q = 0; q < 3; q++) x[q] = in[(7 * i + q) % n];
    for (int q = 0; q < 4; q++) u[q] = in[(7 * i + 3 + q) % n];
    double extra[EXTRA + 1]; for (int q = 0; q <= EXTRA; q++) extra[q] = in[(i + 11 * q) % n];
    double h = 1e-2, lq = in[(i + 5) % n];            // h² below is the value the real kernel lost; lq stands next to it
    accel(x, u, k[0]);
    for (int s = 0; s < steps; s++) {
        const double h2 = h * h;
#pragma unroll
        for (int l = 1; l < 7; l++) {                 // seven-stage explicit step, Nyström form: x + c h u + h² Σ a k
            double X[3], U[4];
            for (int q = 0; q < 4; q++) { double a = 0; for (int m = 0; m < l; m++) a += (0.1 * (m + 1) / l) * k[m][q]; U[q] = u[q] + h * a; }
            for (int q = 0; q < 3; q++) { double a = 0; for (int m = 0; m < l; m++) a += (0.05 * (m + 2) / l) * k[m][1 + q]; X[q] = x[q] + 0.2 * l * h * u[1 + q] + h2 * a; }
            accel(X, U, k[l]);
        }
        double err = 0;
        for (int q = 0; q < 4; q++) { double e = 0; for (int m = 0; m < 7; m++) e += (0.01 * (m - 3)) * k[m][q]; err += e * e; u[q] += h * k[6][q]; }
        for (int q = 0; q < 3; q++) x[q] += h * u[1 + q] + h2 * k[3][1 + q];
        h *= err * h2 < 1e-12 ? 1.5 : 0.5; lq += h;
        for (int q = 0; q <= EXTRA; q++) extra[q] = extra[q] * h2 + k[q % 7][q % 4];
        for (int q = 0; q < 4; q++) k[0][q] = k[6][q];
    }
    double* o = out + (size_t)i * (8 + EXTRA + 1);    // every thread writes slots of its own: no two threads ever touch one word
    for (int q = 0; q < 3; q++) o[q] = x[q];
    for (int q = 0; q < 4; q++) o[3 + q] = u[q] + lq;
    for (int q = 0; q <= EXTRA; q++) o[8 + q] = extra[q];
}

#ifndef __HIP_DEVICE_COMPILE__
#include <cstdio>
#include <cstring>
#include <vector>
#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 2; } } while (0)
static int run(const char* path, const std::vector<double>& in, std::vector<double>& out, int n, int waves, int steps) {
    hipModule_t m; hipFunction_t f;
    OK(hipModuleLoad(&m, path)); OK(hipModuleGetFunction(&f, m, "repro"));
    double *d_in, *d_out;
    const size_t n_out = out.size();
    OK(hipMalloc(&d_in, n * 8)); OK(hipMalloc(&d_out, n_out * 8));
    OK(hipMemcpy(d_in, in.data(), n * 8, hipMemcpyHostToDevice)); OK(hipMemset(d_out, 0, n_out * 8));
    void* args[] = {&d_in, &d_out, &n, &steps};
    OK(hipModuleLaunchKernel(f, waves, 1, 1, 64, 1, 1, 0, nullptr, args, nullptr));
    OK(hipDeviceSynchronize());
    OK(hipMemcpy(out.data(), d_out, n_out * 8, hipMemcpyDeviceToHost));
    OK(hipFree(d_in)); OK(hipFree(d_out)); OK(hipModuleUnload(m));
    return 0;
}
int main(int argc, char** argv) {   // exec_flip_repro as_compiled.hsaco repaired.hsaco
    if (argc < 3) { fprintf(stderr, "usage: %s as_compiled.hsaco repaired.hsaco\n", argv[0]); return 2; }
    const int waves = 64, n = 7 * 64 * waves, steps = 40, n_out = 64 * waves * (8 + EXTRA + 1);
    std::vector<double> in(n), a(n_out), b(n_out), a2(n_out);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; in[i] = 0.5 + 3.0 * (double)(s >> 11) / 9007199254740992.0; }
    if (run(argv[1], in, a, n, waves, steps) || run(argv[1], in, a2, n, waves, steps) || run(argv[2], in, b, n, waves, steps)) return 2;
    int differ = 0, unstable = 0, shown = 0;
    for (int i = 0; i < n_out; i++) {
        unstable += std::memcmp(&a[i], &a2[i], 8) != 0;
        if (std::memcmp(&a[i], &b[i], 8) != 0 && differ++ < 1000000 && shown < 8) { printf("  out[%d]: expected (repaired) %.17g   actual (as compiled) %.17g\n", i, b[i], a[i]); shown++; }
    }
    printf("%d of %d results differ between the kernel as compiled and with its FLOW block repaired; %d differ between two runs of the kernel as compiled\n", differ, n_out, unstable);
    return differ ? 1 : 0;
}
#endif
