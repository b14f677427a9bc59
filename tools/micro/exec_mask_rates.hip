// Does a partially filled wave issue faster?  v_fma_f64 / v_fma_f32 / v_rcp_f64 under EXEC masks that leave whole
// 16-lane quarters of the wave64 empty, one wave per SIMD (the situation of the long-staying rays at the end of a small
// launch: a wave with a handful of live lanes).  If the SIMD skipped the passes of empty quarters, compacting the live
// lanes into the low quarter would shorten those tails; it does not (see profiles/r03/exec_mask_rates.log).
//     hipcc --offload-arch=gfx950 -O3 tools/micro/exec_mask_rates.hip -o exec_mask_rates && ./exec_mask_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ITER 512

template <int OP>
__global__ __launch_bounds__(64) void bench(unsigned long long* out, double seed, unsigned long long lanesA, unsigned long long lanesB) {
    const unsigned long long lanes = (blockIdx.x & 1) ? lanesB : lanesA;   // odd blocks: the other mask (a busy neighbourhood)
    double a[8], b = seed * 1.0000001, c = seed * 0.9999999;
    float fa[8], fb = (float)seed;
    for (int i = 0; i < 8; i++) { a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6; fa[i] = (float)a[i]; }
    unsigned long long t0 = 0, t1 = 0;
    if ((lanes >> threadIdx.x) & 1ull) {   // EXEC = lanes inside
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < ITER; it++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if constexpr (OP == 0) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    REP8(S)
#undef S
                } else if constexpr (OP == 1) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
                    REP8(S)
#undef S
                } else {
#define S(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
                    REP8(S)
#undef S
                }
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    double s = 0; float fs = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i]; }
    if (s == 12345.678 && fs == 1.0f) out[1000000] = 1;
    if (threadIdx.x == (unsigned)__builtin_ctzll(lanes)) out[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, unsigned long long* d, int ncu) {
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0xffffull << 48, 0xfffull, 0xffull, 0xfull, 0x3ull, 0x1ull,
                                        0x0001000100010001ull, 0x0101010101010101ull, 0x1111111111111111ull, 0x5555555555555555ull};
    const char* mname[] = {"all 64", "0-31", "0-15", "48-63", "0-11", "0-7", "0-3", "0-1", "lane 0", "4 spread (every 16th)", "8 spread (every 8th)",
                           "16 spread (every 4th)", "32 spread (every 2nd)"};
    printf("%-10s", name);
    for (int m = 0; m < 13; m++) {
        int blocks = ncu * 4;   // one wave per SIMD
        for (int mixed = 0; mixed < 2; mixed++) {   // every wave under the mask / every other wave with all 64 lanes
            hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(64), 0, 0, d, 1.25, masks[m], mixed ? ~0ull : masks[m]);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (int k = 0; k < blocks; k += 2) avg += (double)h[k]; avg /= blocks / 2;
            if (mixed) printf(" (%.3f beside full waves)", avg / (ITER * 32.0)); else printf("  %s: %.3f", mname[m], avg / (ITER * 32.0));
        }
    }
    printf("   (ticks per instruction, 1 wave per SIMD)\n");
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    unsigned long long* d; hipMalloc(&d, 8 * 1000001);
    printf("%s  %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    run<0>("v_fma_f64", d, p.multiProcessorCount);
    run<1>("v_fma_f32", d, p.multiProcessorCount);
    run<2>("v_rcp_f64", d, p.multiProcessorCount);
    return 0;
}
