// Micro-benchmark: VALU issue cadence of a lone wavefront on gfx950 (what bounds the lane-per-configuration kernels
// at small batch).  Prints cycles per instruction for several instruction kinds at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2000;

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float a, float b) {
    float x[16];
    float2_ p[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; i++) { p[i] = float2_{x[2 * i], x[2 * i + 1]}; d[i] = x[i]; }
    float2_ pa = float2_{a, a}, pb = float2_{b, b};
    double da = a, db = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {           // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if (KIND == 1) {    // 16 dependent v_fma_f32 (one chain)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
        } else if (KIND == 2) {    // 16 v_pk_fma_f32 on 8 independent pairs (2 rounds)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
        } else if (KIND == 3) {    // 16 v_fma_f64 on 8 independent (2 rounds)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
        } else if (KIND == 4) {    // 16 v_fmaak_f32 (32-bit literal in the instruction: 8-byte encoding)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f9e0652" : "+v"(x[i]) : "v"(a));
        } else if (KIND == 5) {    // 16 v_fma_f32 with an SGPR operand
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
        } else if (KIND == 6) {    // 8 x (v_fma_f32 + independent v_mul_f32 + v_add_f32)  -- mixed VOP2/VOP3
#pragma unroll
            for (int i = 0; i < 8; i++) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i + 8]) : "v"(b)); }
        } else if (KIND == 7) {    // 16 dependent v_pk_fma_f32 (one chain)
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pa), "v"(pb));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
int run(const char *name) {
    const int cfgs[4][2] = {{256, 64}, {256, 256}, {256 * 2, 256}, {256 * 4, 256}};  // 1 wave/CU, 1/SIMD, 2/SIMD, 4/SIMD
    printf("%-34s", name);
    for (auto &c : cfgs) {
        int blocks = c[0], threads = c[1];
        float *out; unsigned long long *cyc;
        int nw = blocks * threads / 64;
        CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
        CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * nw));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(nw);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += (double)v;
        // s_memtime counts at a fixed 100 MHz?  report raw ticks per instruction; calibrated by the wall-clock column
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %6.2f tick/inst %7.1f us |", sum / nw / (ITERS * 16.0), ms * 1e3);
        hipFree(out); hipFree(cyc);
    }
    printf("\n");
    return 0;
}

int main() {
    printf("%-34s  %-27s  %-27s  %-27s  %-27s\n", "kind (16 inst x 2000 iters)", "1 wave/CU", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    run<0>("v_fma_f32 x16 independent");
    run<1>("v_fma_f32 x16 dependent chain");
    run<2>("v_pk_fma_f32 x16 (8 indep pairs)");
    run<7>("v_pk_fma_f32 x16 dependent chain");
    run<3>("v_fma_f64 x16 (8 indep)");
    run<4>("v_fmaak_f32 x16 (literal)");
    run<5>("v_fma_f32 x16 sgpr operand");
    run<6>("v_mul_f32/v_add_f32 x16 (VOP2)");
    return 0;
}
