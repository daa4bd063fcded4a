// Micro-benchmark: how many single-wave workgroups of a 512-register kernel run at the same time on an MI355X?
// Each wave spins on a short loop (fits the instruction cache) for a fixed number of instructions; total time vs number
// of blocks shows the number of concurrent slots (time doubles when a second round is needed).
// Variants: REGS = 128 / 256 / 512 registers per lane (forced by clobbering the highest register), LDS bytes per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int REGS>
__global__ __launch_bounds__(64) void spin(float *out, float a, float b, int iters) {
    extern __shared__ float lds[];
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    if (REGS >= 512) asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
    if (REGS >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    else if (REGS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    if (iters < 0) lds[threadIdx.x] = s;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int REGS>
int run(float *d_out, size_t lds) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int blocks : {256, 512, 768, 1024, 1280, 1536, 2048, 4096}) {
        spin<REGS><<<blocks, 64, lds>>>(d_out, 1.0001f, 0.5f, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        spin<REGS><<<blocks, 64, lds>>>(d_out, 1.0001f, 0.5f, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("regs %3d  lds %6zu B  blocks %5d  %8.1f us\n", REGS, lds, blocks, ms * 1e3);
    }
    return 0;
}

int main() {
    float *d_out; CHECK(hipMalloc(&d_out, sizeof(float) * 64 * 8192));
    if (run<512>(d_out, 0)) return 1;
    if (run<512>(d_out, 15360)) return 1;
    if (run<512>(d_out, 38912)) return 1;
    if (run<256>(d_out, 0)) return 1;
    if (run<128>(d_out, 0)) return 1;
    return 0;
}
