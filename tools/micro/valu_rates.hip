// Issue cost of the VALU instructions the integrate kernel is made of, on gfx950: cycles per wave64 instruction on one
// SIMD, measured with s_memtime around long unrolled runs of INDEPENDENT instructions (8 accumulator chains), at 1, 2
// and 3 waves per SIMD.    hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ITER 512

template <int OP>
__global__ __launch_bounds__(64) void bench(unsigned long long* out, double seed) {
    double a[8], b = seed * 1.0000001, c = seed * 0.9999999;
    float fa[8], fb = (float)seed, fc = fb * 1.5f;
    unsigned long long mask = __ballot(threadIdx.x & 1), m2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned ia[8] = {1, 2, 3, 4, 5, 6, 7, 8}, ib = threadIdx.x;
    for (int i = 0; i < 8; i++) { a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6; fa[i] = (float)a[i]; }
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if constexpr (OP == 0) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(S)
#undef S
            } else if constexpr (OP == 1) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(S)
#undef S
            } else if constexpr (OP == 2) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(S)
#undef S
            } else if constexpr (OP == 3) {
#define S(i) asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(S)
#undef S
            } else if constexpr (OP == 4) {
#define S(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 5) {
#define S(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 6) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(fa[i]) : "v"(a[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 7) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
                REP8(S)
#undef S
            } else if constexpr (OP == 8) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(fa[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 9) {
#define S(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(S)
#undef S
            } else if constexpr (OP == 10) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(fa[i]) : "v"(fb) : );
                REP8(S)
#undef S
            } else if constexpr (OP == 11) {
#define S(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                REP8(S)
#undef S
            } else if constexpr (OP == 12) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "+v"(fa[i]) : "v"(fb));
                REP8(S)
#undef S
            } else if constexpr (OP == 13) {
#define S(i) asm volatile("v_log_f32 %0, %0" : "+v"(fa[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 14) {
#define S(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(S)
#undef S
            } else if constexpr (OP == 16) {
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(fa[i]) : "v"(fb), "s"(mask));
                REP8(S)
#undef S
            } else if constexpr (OP == 17) {
#define S(i) asm volatile("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(m2[i]) : "v"(a[i]), "v"(b));
                REP8(S)
#undef S
            } else if constexpr (OP == 18) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(a[i]) : "v"(fa[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 19) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(fa[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 20) {
#define S(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(fa[i]));
                REP8(S)
#undef S
            } else if constexpr (OP == 21) {
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
                REP8(S)
#undef S
            } else if constexpr (OP == 22) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
                REP8(S)
#undef S
            } else if constexpr (OP == 23) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                REP8(S)
#undef S
            } else if constexpr (OP == 24) {   // e32 cndmask, dst != src, vcc written once before the loop
#define S(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(fa[i]) : "v"(fb), "v"(fc) : );
                REP8(S)
#undef S
            } else if constexpr (OP == 25) {   // compare + select pairs as the compiler emits them
#define S(i) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(fa[i]) : "v"(fb), "v"(fc) : "vcc");
                REP8(S)
#undef S
            } else if constexpr (OP == 26) {   // compare into an SGPR pair + e64 select
#define S(i) asm volatile("v_cmp_gt_f32_e64 %3, %1, %2\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(fa[i]) : "v"(fb), "v"(fc), "s"(m2[i]) : );
                REP8(S)
#undef S
            } else if constexpr (OP == 15) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                REP8(S)
#undef S
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; float fs = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i]; }
    unsigned long long ms = 0; unsigned is_ = 0;
    for (int i = 0; i < 8; i++) { ms += m2[i]; is_ += ia[i]; }
    if (s == 12345.678 && fs == 1.0f && ms == 77 && is_ == 99) out[1000000] = 1;   // keep the chains alive
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, unsigned long long* d, int ncu) {
    printf("%-16s", name);
    for (int wps = 1; wps <= 3; wps++) {  // waves per SIMD (4 SIMDs per CU)
        int blocks = ncu * 4 * wps;
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(64), 0, 0, d, 1.25);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
        // s_memtime / readcyclecounter ticks at a constant 100 MHz on this part; report both raw and per instruction
        printf("  %dw: %8.1f ticks -> %.4f ticks/inst/wave", wps, avg, avg / (ITER * 32.0) / wps);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount;
    printf("%s  %d CUs  clock %d kHz  (readcyclecounter rate: see ratio to v_fma_f64)\n", p.gcnArchName, ncu, p.clockRate);
    unsigned long long* d; hipMalloc(&d, 8 * 1000001);
    run<0>("v_fma_f64", d, ncu); run<14>("v_fmac_f64", d, ncu); run<1>("v_mul_f64", d, ncu); run<2>("v_add_f64", d, ncu); run<9>("v_max_f64", d, ncu);
    run<3>("v_mov_b64", d, ncu); run<12>("v_mov_b32", d, ncu); run<10>("v_cndmask_b32", d, ncu); run<11>("v_cmp_gt_f64", d, ncu);
    run<4>("v_rcp_f64", d, ncu); run<5>("v_rsq_f64", d, ncu); run<6>("v_cvt_f32_f64", d, ncu);
    run<16>("v_cndmask e64", d, ncu); run<17>("v_cmp_f64 e64", d, ncu); run<18>("v_cvt_f64_f32", d, ncu); run<19>("v_exp_f32", d, ncu);
    run<20>("v_sqrt_f32", d, ncu); run<21>("v_max_f32", d, ncu); run<22>("v_add_u32", d, ncu); run<23>("v_lshl_add_u64", d, ncu);
    run<24>("cndmask e32 d!=s", d, ncu); run<25>("cmp+cndmask vcc", d, ncu); run<26>("cmp+cndmask sgpr", d, ncu);
    run<7>("v_fma_f32", d, ncu); run<15>("v_pk_fma_f32", d, ncu); run<8>("v_rcp_f32", d, ncu); run<13>("v_log_f32", d, ncu);
    return 0;
}
