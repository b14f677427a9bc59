// How do gfx950's SQ counters count PACKED f32 arithmetic?  One wave per kernel issues exactly 4096 instructions of one kind;
//     rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU -- ./pk_counter_probe
// reads them per kernel (profiles/r05/pk_counter_probe.log).  bench.py's live roofline of the packed Float32 kernel (two rays per
// lane, v_pk_fma_f32) depends on the answer: if a packed instruction is counted ONCE by SQ_INSTS_VALU_FMA_F32 the per-kind counters
// see half its flops, and SQ_INSTS_VALU_FLOPS_FP32 is the counter to read.
//     hipcc -w --offload-arch=gfx950 -O3 tools/micro/pk_counter_probe.hip -o tools/micro/pk_counter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define KERNEL(name, T, ASM)                                                                     \
    extern "C" __global__ __launch_bounds__(64) void name(float* out, float seed) {              \
        T a = T(seed + threadIdx.x), b = T(1.0000001f), c = T(1e-7f);                            \
        for (int it = 0; it < 512; it++) {                                                       \
            asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM : "+v"(a) : "v"(b), "v"(c)); \
        }                                                                                        \
        out[threadIdx.x] = sum_of(a);                                                            \
    }
static __device__ float sum_of(float a) { return a; }
static __device__ float sum_of(float2_t a) { return a.x + a.y; }
KERNEL(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_pk_fma_f32, float2_t, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL(k_mul_f32, float, "v_mul_f32 %0, %0, %1")
KERNEL(k_pk_mul_f32, float2_t, "v_pk_mul_f32 %0, %0, %1")
KERNEL(k_add_f32, float, "v_add_f32 %0, %0, %2")
KERNEL(k_pk_add_f32, float2_t, "v_pk_add_f32 %0, %0, %2")
int main() {
    float* d;
    (void)hipMalloc(&d, 256);
    hipLaunchKernelGGL(k_fma_f32, dim3(1), dim3(64), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(k_pk_fma_f32, dim3(1), dim3(64), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(k_mul_f32, dim3(1), dim3(64), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(k_pk_mul_f32, dim3(1), dim3(64), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(k_add_f32, dim3(1), dim3(64), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(k_pk_add_f32, dim3(1), dim3(64), 0, 0, d, 1.0f);
    (void)hipDeviceSynchronize();
    printf("six kernels, one wave each, 4096 instructions of one kind per wave\n");
    return 0;
}
