// What does a hand-over of ray records between workgroups on DIFFERENT XCDs cost INSIDE one kernel on gfx950?
// (DESIGN §6 "what would overlap inside one small frame": a NEAR kernel consuming the FAR pass's hand-overs while the FAR pass is
//  still running would be an in-kernel producer / consumer across XCDs.  Each XCD has its own L2 and coarse-grained device memory is
//  not coherent between them within a kernel: this measures what making it coherent costs, and shows what happens without.)
//
//     hipcc -w --offload-arch=gfx950 -O3 tools/micro/xcd_handover.hip -o tools/micro/xcd_handover && tools/micro/xcd_handover
//
// 1024 one-wave workgroups = 512 pairs: workgroup 2p PRODUCES, workgroup 2p + 1 CONSUMES (consecutive workgroups go to consecutive
// XCDs, so the two of a pair sit on different ones — each reports its XCC_ID and the host counts).  Per iteration the producer's 64
// lanes write one 128-byte record each (the pipeline's 16-scalar hand-over record), lane 0 publishes a sequence number, and the
// consumer waits for it (bounded spin), reads the 64 records and verifies every scalar.  Between hand-overs a producer lane also
// writes `noise` plain 8-byte stores to lines of its own (the event records and meta words a FAR wave writes anyway): dirty lines
// in the XCD's L2 that a release fence has to write back.
//   mode 0  plain stores, plain loads, relaxed flag           -> no coherence action at all: how many stale scalars are read?
//   mode 1  plain stores + __threadfence() | flag | __threadfence() + plain loads     -> release / acquire at agent scope (L2 write-back / invalidate)
//   mode 2  agent-scope atomic stores (write-through), s_waitcnt, flag | agent-scope atomic loads      -> no fence, every access past the L2
// Output per (mode, noise): ns per hand-over as the producer sees it, as the consumer sees it, stale scalars, spins that gave up.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int REC = 16;                    // doubles per record
constexpr int NOISE_LINES = 8192;          // 128-byte lines of noise space per producer (1 MiB)
constexpr unsigned MAXSPIN = 1u << 20;     // bounded: a consumer that never sees its flag gives up and says so

__device__ inline double expect(unsigned p, unsigned i, unsigned lane, unsigned q) { return (double)(((p * 131u + i) * 64u + lane) * 16u + q) + 0.5; }

template <int MODE>
__global__ __launch_bounds__(64) void handover(double* rec, unsigned* flag, unsigned long long* stats, double* noise, unsigned iters, unsigned nnoise) {
    const unsigned b = blockIdx.x, p = b >> 1, lane = threadIdx.x;
    const bool producer = (b & 1u) == 0u;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long stale = 0, gaveup = 0;
    const unsigned long long t0 = wall_clock64();
    for (unsigned i = 0; i < iters; i++) {
        double* r = rec + (((unsigned long long)p * iters + i) * 64ull + lane) * REC;
        if (producer) {
            double* nz = noise + (unsigned long long)p * NOISE_LINES * 16ull;
            for (unsigned s = 0; s < nnoise; s++) nz[(((i * nnoise + s) * 64u + lane) % NOISE_LINES) * 16u] = (double)(i + s);
            if (MODE == 2) {
#pragma unroll
                for (unsigned q = 0; q < REC; q++) __hip_atomic_store(r + q, expect(p, i, lane, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);           // every store of the wave has been acknowledged
            } else {
#pragma unroll
                for (unsigned q = 0; q < REC; q++) r[q] = expect(p, i, lane, q);
                if (MODE == 1) __threadfence();          // release, agent scope: the XCD's dirty L2 lines go out
            }
            if (lane == 0) __hip_atomic_store(flag + p, i + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(flag + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i + 1u && spins < MAXSPIN) { spins++; __builtin_amdgcn_s_sleep(2); }
            if (spins >= MAXSPIN) { gaveup++; continue; }
            if (MODE == 1) __threadfence();              // acquire, agent scope: non-local L2 lines are dropped
#pragma unroll
            for (unsigned q = 0; q < REC; q++) {
                const double v = MODE == 2 ? __hip_atomic_load(r + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : r[q];
                stale += v != expect(p, i, lane, q);
            }
        }
    }
    const unsigned long long dt = wall_clock64() - t0;
    for (int o = 32; o; o >>= 1) { stale += __shfl_xor(stale, o); gaveup = gaveup > __shfl_xor(gaveup, o) ? gaveup : __shfl_xor(gaveup, o); }
    if (lane == 0) { stats[4ull * b] = dt; stats[4ull * b + 1] = stale; stats[4ull * b + 2] = gaveup; stats[4ull * b + 3] = xcc; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const unsigned pairs = 512, iters = 64;
    double *rec, *noise;
    unsigned* flag;
    unsigned long long* stats;
    const size_t rec_bytes = (size_t)pairs * iters * 64 * REC * sizeof(double);
    CK(hipMalloc(&rec, rec_bytes));
    CK(hipMalloc(&noise, (size_t)pairs * NOISE_LINES * 128));
    CK(hipMalloc(&flag, pairs * sizeof(unsigned)));
    CK(hipMalloc(&stats, 2 * pairs * 4 * sizeof(unsigned long long)));
    int rate_khz = 100000;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double ns_per_tick = 1e6 / rate_khz;
    std::vector<unsigned long long> h(2 * pairs * 4);
    printf("512 producer / consumer pairs of one-wave workgroups, %u hand-overs of 64 x 128-byte records each; wall clock %d kHz\n", iters, rate_khz);
    for (unsigned nnoise : {0u, 8u, 64u})
        for (int mode = 0; mode < 3; mode++) {
            CK(hipMemset(rec, 0, rec_bytes));
            CK(hipMemset(flag, 0, pairs * sizeof(unsigned)));
            CK(hipDeviceSynchronize());
            if (mode == 0) hipLaunchKernelGGL(handover<0>, dim3(2 * pairs), dim3(64), 0, 0, rec, flag, stats, noise, iters, nnoise);
            if (mode == 1) hipLaunchKernelGGL(handover<1>, dim3(2 * pairs), dim3(64), 0, 0, rec, flag, stats, noise, iters, nnoise);
            if (mode == 2) hipLaunchKernelGGL(handover<2>, dim3(2 * pairs), dim3(64), 0, 0, rec, flag, stats, noise, iters, nnoise);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double tp = 0, tc = 0;
            unsigned long long stale = 0, gaveup = 0, cross = 0;
            for (unsigned p = 0; p < pairs; p++) {
                tp += (double)h[8 * p] * ns_per_tick / iters;
                tc += (double)h[8 * p + 4] * ns_per_tick / iters;
                stale += h[8 * p + 5];
                gaveup += h[8 * p + 6];
                cross += h[8 * p + 3] != h[8 * p + 7];
            }
            static const char* const NAME[] = {"plain stores / loads, no fence       ", "plain + __threadfence() both sides   ", "agent-scope atomic stores / loads    "};
            printf("noise %2u stores/lane  mode %d %s: producer %8.0f ns, consumer %8.0f ns per hand-over; %llu of %llu scalars stale; %llu spins gave up; %llu of %u pairs span two XCDs\n",
                   nnoise, mode, NAME[mode], tp / pairs, tc / pairs, stale, (unsigned long long)pairs * iters * 64 * REC, gaveup, cross, pairs);
        }
    return 0;
}
