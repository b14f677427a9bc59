// Can a few register-heavy "side" waves (launched FIRST, on a second stream, polling a flag) share the chip with a
// machine-filling persistent kernel launched after them — and what does the big kernel lose?  (The question behind running
// the NEAR pass's long stayers beside the FAR pass instead of after it)
//   side kernel:  G waves x 64 lanes, ~250 VGPRs (one SIMD slot each; a 144-register wave still fits beside it), polls
//                 a word every ~2 us until the main kernel's last wave sets it (or 20 ms pass: it can never hang)
//   main kernel:  3 waves per SIMD x 1024 SIMDs, ~144 VGPRs, persistent: chunks of dependent-free f64 FMA work popped from a queue
//     hipcc --offload-arch=gfx950 -O3 tools/micro/coreside_probe.hip -o coreside_probe && ./coreside_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long realtime() { return __builtin_readcyclecounter(); }

template <int NREG>
__device__ __forceinline__ void fma_work(double (&a)[NREG], double b, double c, int iters) {
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NREG; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
}

// ~250 VGPRs: 120 live doubles
__global__ __launch_bounds__(64, 2) void side_kernel(unsigned* flag, unsigned long long* out, double seed, unsigned long long max_ticks, int work_iters) {
    double a[120];
    for (int i = 0; i < 120; i++) a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0, seen = 0;
    while (true) {
        fma_work<120>(a, 1.0000001, 1e-9, work_iters);   // the side waves' own (small) work between polls
        polls++;
        seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (seen || wall_clock64() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
    double s = 0;
    for (int i = 0; i < 120; i++) s += a[i];
    if (s == 12345.678) out[100000] = 1;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = wall_clock64() - t0; out[2 * blockIdx.x + 1] = ((unsigned long long)seen << 32) | polls; }
}

// ~144 VGPRs: 64 live doubles
// (persistent, like the integrate passes: every wave pops chunks of work from one queue until it is empty, so a wave that
//  starts late — because a side wave holds its slot — simply finds less to do)
__global__ __launch_bounds__(64, 3) void main_kernel(unsigned* flag, unsigned* done_ctr, unsigned grid, double seed, int iters) {
    double a[64];
    for (int i = 0; i < 64; i++) a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
    const unsigned nchunks = grid * 40u;
    while (true) {
        unsigned c = 0;
        if (threadIdx.x == 0) c = atomicAdd(done_ctr + 1, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= nchunks) break;
        fma_work<64>(a, 1.0000001, 1e-9, iters / 40);
    }
    double s = 0;
    for (int i = 0; i < 64; i++) s += a[i];
    if (s == 12345.678) flag[1] = 1;
    if (threadIdx.x == 0) {
        const unsigned k = atomicAdd(done_ctr, 1u);
        if (k + 1 == grid) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // last wave out
    }
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    unsigned* flag; unsigned* ctr; unsigned long long* out;
    CHECK(hipMalloc(&flag, 64)); CHECK(hipMalloc(&ctr, 64)); CHECK(hipMalloc(&out, 8 * 100001));
    hipStream_t s_main, s_side; CHECK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = ncu * 4 * 3;
    const int iters = 6000;   // ~ several ms of main-kernel work
    printf("%s  %d CUs; main grid %u waves (3 per SIMD)\n", p.gcnArchName, ncu, grid);
    for (int side_waves : {0, 32, 64, 128, 256}) {
        for (int order = 0; order < (side_waves ? 2 : 1); order++) {   // 0: side first (the intended order), 1: main first
            float best = 1e9f; double side_ms = 0; unsigned long long seen_polls = 0;
            for (int rep = 0; rep < 6; rep++) {
                CHECK(hipMemsetAsync(flag, 0, 64, s_main)); CHECK(hipMemsetAsync(ctr, 0, 64, s_main));
                CHECK(hipStreamSynchronize(s_main));
                if (side_waves && order == 0) hipLaunchKernelGGL(side_kernel, dim3(side_waves), dim3(64), 0, s_side, flag, out, 1.25, 2000000ull, 20);
                CHECK(hipEventRecord(e0, s_main));
                hipLaunchKernelGGL(main_kernel, dim3(grid), dim3(64), 0, s_main, flag, ctr, grid, 1.25, iters);
                CHECK(hipEventRecord(e1, s_main));
                if (side_waves && order == 1) hipLaunchKernelGGL(side_kernel, dim3(side_waves), dim3(64), 0, s_side, flag, out, 1.25, 2000000ull, 20);
                CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                if (side_waves) {
                    std::vector<unsigned long long> h(2 * side_waves);
                    CHECK(hipMemcpy(h.data(), out, 16 * side_waves, hipMemcpyDeviceToHost));
                    double avg = 0; for (int k = 0; k < side_waves; k++) avg += (double)h[2 * k];
                    side_ms = avg / side_waves * 1e-5;   // wall_clock64: 100 MHz
                    seen_polls = h[1];
                }
            }
            if (!side_waves) printf("main alone: %.3f ms\n", best);
            else printf("side %3d waves, %s: main %.3f ms; side waves lived %.3f ms on average (flag seen %llu, polls %llu)\n", side_waves,
                        order == 0 ? "side launched first" : "main launched first", best, side_ms, seen_polls >> 32, seen_polls & 0xffffffffull);
        }
    }
    return 0;
}
