// Micro-benchmark: does a straight-line instruction stream larger than the instruction cache slow a wavefront down?
// Each kernel executes TOTAL independent v_fma_f32 (8-byte VOP3 encodings, 16 accumulators), either as one straight-line
// block of BODY instructions repeated TOTAL/BODY times in a loop (BODY*8 bytes of code: 4 KB .. 64 KB) or executed once
// (128 KB .. 1 MB).  Reports ns and
// cycles per instruction per wave with 1, 2 and 4 waves per SIMD (grid = waves/SIMD * 1024 single-wave blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define XSTR(s) STR(s)
#define STR(s) #s

template <int BODY16>   // straight-line block = BODY16 * 16 instructions
__global__ __launch_bounds__(64) void k(float *out, float a, float b, int reps, int desync) {
    float x[16];
    if (desync) {   // start the waves of a CU at different times: they then stream different parts of the code
        const unsigned d = ((blockIdx.x * 2654435761u) >> 13) % 8u;
        for (unsigned i = 0; i < d * (unsigned)desync; i++) __builtin_amdgcn_s_sleep(127);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < (BODY16 > 512 ? 1 : reps); it++) {
        asm volatile(".rept %18\n"
                     "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                     "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                     "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                     "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                     ".endr\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                       "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                     : "v"(a), "v"(b), "n"(BODY16));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BODY16>
int run(float *d_out, int total16, int desync = 0) {
    const int reps = BODY16 > 512 ? 1 : total16 / BODY16;      // bodies > 64 KB: executed once (s_cbranch reaches +-128 KB)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 1024 * wps;
        k<BODY16><<<blocks, 64>>>(d_out, 1.0001f, 0.5f, reps, desync);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<BODY16><<<blocks, 64>>>(d_out, 1.0001f, 0.5f, reps, desync);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ninst = 16.0 * BODY16 * reps;
        printf("desync %d  code %7d B  waves/SIMD %d  %8.1f us  %.3f ns/instr/wave (%.2f cycles @2.4GHz)  SIMD issue interval %.2f cycles\n", desync, BODY16 * 16 * 8, wps, ms * 1e3,
               ms * 1e6 / ninst, ms * 1e6 / ninst * 2.4, ms * 1e6 / ninst * 2.4 / wps);
    }
    return 0;
}

int main() {
    float *d_out; CHECK(hipMalloc(&d_out, sizeof(float) * 64 * 4096));
    const int total16 = 8192;     // 131072 instructions per wave
    if (run<32>(d_out, total16)) return 1;      //   4 KB
    if (run<256>(d_out, total16)) return 1;     //  32 KB
    if (run<512>(d_out, total16)) return 1;     //  64 KB
    if (run<1024>(d_out, total16)) return 1;    // 128 KB
    if (run<4096>(d_out, total16)) return 1;    // 512 KB
    if (run<8192>(d_out, total16)) return 1;    //   1 MB, executed once
    // desynchronised: wave w of a CU starts (hash % 8) * 6 * 3.4 us late (64 KB of code is ~17 us of execution)
    if (run<256>(d_out, total16, 6)) return 1;
    if (run<4096>(d_out, total16, 6)) return 1;
    if (run<8192>(d_out, total16, 6)) return 1;
    return 0;
}
