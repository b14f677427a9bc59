// What can gfx950 issue for the instruction MIX of one wave-step of the a = 0 FAR pass when no instruction ever waits for an
// operand?  mix_replay_body.inc (tools/micro/gen_mix_replay.py) is that mix — 812 VALU instructions, 658 of them f64
// FMA/MUL/ADD carrying 1087 flop per lane — as independent instructions on 8 accumulator chains.  Run chip-wide at 1-4 waves
// per SIMD for ~50 ms of wall time each (so the clock is what the power limit gives under this load), it yields "wave-steps
// per second" — the ceiling of THIS mix — to set beside the real kernel's 4.07e10 step attempts/s = 6.4e8 wave-steps/s.
//     python tools/micro/gen_mix_replay.py && hipcc --offload-arch=gfx950 -O3 tools/micro/mix_replay.hip -o mix_replay && ./mix_replay
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "mix_replay_counts.inc"

template <bool PURE_FMA>
__global__ __launch_bounds__(64) void replay(unsigned long long* out, double seed, int iters) {
    double a[8], b = seed * 1.0000001, c = seed * 0.9999999;
    float fa[8], fb = (float)seed, fc = fb * 1.5f;
    unsigned long long mask = __ballot(threadIdx.x & 1), m2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned ia[8] = {1, 2, 3, 4, 5, 6, 7, 8}, ib = threadIdx.x;
    for (int i = 0; i < 8; i++) { a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6; fa[i] = (float)a[i]; }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if constexpr (PURE_FMA) {
#include "mix_replay_fma.inc"
        } else {
#include "mix_replay_body.inc"
        }
    }
    double s = 0; float fs = 0;
    unsigned long long ms = 0; unsigned is_ = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i]; ms += m2[i]; is_ += ia[i]; }
    if (s == 12345.678 && fs == 1.0f && ms == 77 && is_ == 99) out[0] = 1;   // keep the chains alive
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    unsigned long long* d; (void)hipMalloc(&d, 64);
    printf("%s  %d CUs\n", p.gcnArchName, ncu);
    for (int pure = 0; pure < 2; pure++) {
        auto kern = pure ? replay<true> : replay<false>;
        const double flop_per_lane = pure ? 2.0 * MIX_TOTAL : (double)MIX_FLOP;
        if (pure) printf("-- %d independent v_fma_f64 per iteration: the attainable FMA peak under this chip's power limit\n", MIX_TOTAL);
        else printf("-- the instruction mix of %s (%d VALU per wave-step, %d f64 flop per lane)\n", MIX_NAME, MIX_TOTAL, MIX_FLOP);
        for (int wps = 1; wps <= 4; wps++) {
            const int blocks = ncu * 4 * wps;
            const int iters = (int)(36000.0 * 812 / MIX_TOTAL) / wps;       // ~50-60 ms per launch
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1.25, 2000);   // warm
            (void)hipDeviceSynchronize();
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1.25, iters);
                (void)hipDeviceSynchronize();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                best = dt < best ? dt : best;
            }
            const double wave_steps = (double)blocks * iters;
            printf("%d wave(s)/SIMD: %.1f ms, %.3e iterations/s (x64 lanes = %.3e step attempts/s); %.1f TFLOP/s f64 (%.3f of 78.6)\n",
                   wps, best * 1e3, wave_steps / best, wave_steps / best * 64, wave_steps * 64 * flop_per_lane / best / 1e12,
                   wave_steps * 64 * flop_per_lane / best / 78.6e12);
        }
    }
    return 0;
}
