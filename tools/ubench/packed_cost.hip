// Micro-benchmark (GPU box): what would carrying the (d/dq, d/dqd) pair of the gradient recursions in v_pk_fma_f32 cost a lone
// wavefront?  gfx950's VOP3P encoding has no literal operand, so a model constant of a packed multiply-add must come from an SGPR
// (s_mov_b32 literal first) and a per-lane scalar multiplier (an entry of X(q)) is broadcast with op_sel.  Cycles per
// instruction GROUP at one wave per SIMD, measured with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2000;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float a, float b) {
    float x[16]; float2_ p[8];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = float2_{x[2 * i], x[2 * i + 1]};
    float2_ pa = float2_{a, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0) {            // 16 scalar multiply-adds by a literal constant (what the kernels do today): 16 results
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f9e0652" : "+v"(x[i]) : "v"(a));
        } else if (KIND == 1) {     // 8 x (s_mov_b32 literal + v_pk_fma_f32 with the SGPR broadcast to both halves): 16 results
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("s_mov_b32 s20, 0x3f9e0652\n\tv_pk_fma_f32 %0, %0, s[20:21], %1 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa) : "s20", "s21");
        } else if (KIND == 2) {     // 8 x v_pk_fma_f32 with a per-lane scalar (low half of a register pair) broadcast: 16 results
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pa), "v"(p[(i + 1) & 7]));
        } else if (KIND == 3) {     // 16 scalar v_fma_f32 on registers: 16 results
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        } else if (KIND == 4) {     // 8 x plain v_pk_fma_f32: 16 results
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(p[(i + 1) & 7]));
        } else if (KIND == 5) {     // 16 s_mov_b32 alone
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("s_mov_b32 s20, 0x3f9e0652" ::: "s20");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
int run(const char *name, int per_iter) {
    const int blocks = 256, threads = 256;       // one wave per SIMD
    float *out; unsigned long long *cyc;
    const int nw = blocks * threads / 64;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * threads)); CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * nw));
    for (int r = 0; r < 50; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : h) sum += (double)v;
    printf("%-72s %6.2f cycles per group of %d instructions = %5.2f cycles per result\n", name, sum / nw / ITERS, per_iter, sum / nw / ITERS / 16.0);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    run<0>("16 x v_fmaak_f32 (scalar, literal constant)", 16);
    run<1>("8 x (s_mov_b32 literal + v_pk_fma_f32 sgpr broadcast)", 16);
    run<2>("8 x v_pk_fma_f32, per-lane scalar broadcast with op_sel", 8);
    run<3>("16 x v_fma_f32 (scalar, registers)", 16);
    run<4>("8 x v_pk_fma_f32 (registers)", 8);
    run<5>("16 x s_mov_b32 literal", 16);
    return 0;
}
