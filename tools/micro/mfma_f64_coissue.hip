// Does v_mfma_f64_16x16x4_f64 issue BESIDE v_fma_f64 on gfx950, or do the two share the SIMD's double-precision pipe?
// (BASELINE.json north_star: "MFMA only if the Christoffel 4x4x4 contraction proves a true dense hotspot"; round-4 review:
//  DESIGN §3's "nothing GEMM-shaped to put on MFMA" was a sentence, not a measurement.)
//
//     hipcc -w --offload-arch=gfx950 -O3 tools/micro/mfma_f64_coissue.hip -o tools/micro/mfma_f64_coissue && tools/micro/mfma_f64_coissue
//
// Every wave runs INDEPENDENT instruction chains (nothing ever waits for an operand) until a wall-clock deadline and counts what it
// issued; a workgroup is 8 waves = 2 per SIMD of its CU (wave w of a workgroup lands on SIMD w mod 4), 256 workgroups = the chip.
//   mode 0  both waves of every SIMD: v_fma_f64                               -> the vector peak as this chip clocks it
//   mode 1  both waves of every SIMD: v_mfma_f64_16x16x4_f64                  -> the matrix peak
//   mode 2  one wave of every SIMD v_fma_f64, the other v_mfma_f64             -> do the two add up?  (separate pipes: ~ sum; one pipe: ~ max)
//   mode 3  every wave alternates 1 MFMA : 4 FMA inside ONE instruction stream -> the same question for a single wave (what a fused step would do)
//   mode 4  one wave of every SIMD v_fma_f64, the other idle                    -> what ONE wave per SIMD issues (reference for mode 2's FMA half)
//   mode 5  one wave of every SIMD v_mfma_f64, the other idle
// Output: TFLOP/s per role and combined.  A v_fma_f64 is 64 lanes x 2 flop = 128 flop per wave-instruction, a
// v_mfma_f64_16x16x4_f64 is 16 x 16 x 4 x 2 = 2048.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define FMA8                                                                   \
    asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"       \
                 "v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"       \
                 "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"       \
                 "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"       \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c))
#define MFMA1(i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[i], 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* counts, double* sink, unsigned long long ticks, double seed) {
    const unsigned wave = threadIdx.x >> 6;
    const bool second = wave >= 4;   // the SIMD's second wave
    double a[8];
    double4_t acc[4];
    const double b = seed * 1.0000001, c = seed * 0.9999999;
    for (int i = 0; i < 8; i++) a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
    for (int i = 0; i < 4; i++) acc[i] = double4_t{seed, seed * 0.5, seed * 0.25, seed * 0.125};
    unsigned long long n_fma = 0, n_mfma = 0;
    const bool do_fma = MODE == 0 || (MODE == 2 && !second) || (MODE == 4 && !second);
    const bool do_mfma = MODE == 1 || (MODE == 2 && second) || (MODE == 5 && !second);
    const unsigned long long t_end = wall_clock64() + ticks;
    if (MODE == 3) {
        do {
#pragma unroll
            for (int r = 0; r < 8; r++) {   // 1 MFMA : 4 FMA, eight times (two MFMA per accumulator: back-to-back dependent MFMAs are 4 apart)
                MFMA1(r & 3);
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(a[(4 * r) & 7]), "+v"(a[(4 * r + 1) & 7]), "+v"(a[(4 * r + 2) & 7]), "+v"(a[(4 * r + 3) & 7]) : "v"(b), "v"(c));
            }
            n_fma += 32; n_mfma += 8;
        } while (wall_clock64() < t_end);
    } else if (do_fma) {
        do {
#pragma unroll
            for (int r = 0; r < 16; r++) FMA8;
            n_fma += 128;
        } while (wall_clock64() < t_end);
    } else if (do_mfma) {
        do {
#pragma unroll
            for (int r = 0; r < 8; r++) { MFMA1(0); MFMA1(1); MFMA1(2); MFMA1(3); }
            n_mfma += 32;
        } while (wall_clock64() < t_end);
    }
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long w = (unsigned long long)blockIdx.x * 8 + wave;
        counts[2 * w] = n_fma;
        counts[2 * w + 1] = n_mfma;
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678) sink[0] = s;   // keep the chains alive
}

template <int MODE>
static void run(const char* what, int blocks, double ms) {
    unsigned long long* d_counts;
    double* d_sink;
    const size_t n = (size_t)blocks * 8 * 2;
    hipMalloc(&d_counts, n * sizeof(unsigned long long));
    hipMalloc(&d_sink, sizeof(double));
    hipMemset(d_counts, 0, n * sizeof(unsigned long long));
    int rate_khz = 100000;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks = (unsigned long long)(ms * rate_khz);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, d_counts, d_sink, ticks / 10, 1.0);   // warm the clocks
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, d_counts, d_sink, ticks, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float el = 0;
    hipEventElapsedTime(&el, e0, e1);
    std::vector<unsigned long long> h(n);
    hipMemcpy(h.data(), d_counts, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double f = 0, m = 0;
    for (size_t w = 0; w < n / 2; w++) { f += (double)h[2 * w]; m += (double)h[2 * w + 1]; }
    const double tf_fma = f * 128.0 / (el * 1e-3) * 1e-12, tf_mfma = m * 2048.0 / (el * 1e-3) * 1e-12;
    printf("%-58s %7.2f ms   v_fma_f64 %6.2f TFLOP/s   v_mfma_f64 %6.2f TFLOP/s   sum %6.2f\n", what, el, tf_fma, tf_mfma, tf_fma + tf_mfma);
    hipFree(d_counts); hipFree(d_sink);
}

int main(int argc, char** argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 40.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;   // one 8-wave workgroup per CU: 2 waves per SIMD
    printf("%s, %d CUs, %d workgroups of 8 waves (2 per SIMD), window %.0f ms\n", p.gcnArchName, p.multiProcessorCount, blocks, ms);
    run<0>("mode 0: 2 x v_fma_f64 per SIMD", blocks, ms);
    run<1>("mode 1: 2 x v_mfma_f64_16x16x4_f64 per SIMD", blocks, ms);
    run<2>("mode 2: 1 x v_fma_f64 wave + 1 x v_mfma_f64 wave per SIMD", blocks, ms);
    run<3>("mode 3: every wave 1 MFMA : 4 FMA in one stream", blocks, ms);
    run<4>("mode 4: 1 x v_fma_f64 wave per SIMD alone", blocks, ms);
    run<5>("mode 5: 1 x v_mfma_f64 wave per SIMD alone", blocks, ms);
    return 0;
}
