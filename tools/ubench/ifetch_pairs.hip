// Micro-benchmark: instruction-fetch bandwidth of co-resident waves by ENCODING SIZE.  Straight-line streams of independent
// multiply-adds (16 accumulators) in four encodings -- VOP2 (4 bytes: v_fmac_f32), VOP3 (8 bytes: v_fma_f32), VOP2 + 32-bit
// literal (8 bytes: v_fmaak_f32; gfx950 has no literal in VOP3, so 8 bytes is the largest arithmetic encoding) and the mix of the
// generated iiwa-7 kernels (54 % 4-byte, 46 % 8-byte: 5.8 bytes per instruction) -- as a 4 KB loop, a 64 KB loop (the size of the
// unsplit forward-dynamics-gradient kernel) and a 256 KB block executed once, with 1, 2 and 4 waves per SIMD.  Prints cycles per
// instruction per wave (s_memtime around the stream, median over waves), the SIMD's issue interval and the bytes of code each
// SIMD consumes per cycle.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OPS16(a, b, c, d) a(0) b(1) c(2) d(3) a(4) b(5) c(6) d(7) a(8) b(9) c(10) d(11) a(12) b(13) c(14) d(15)
#define FMAC(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define FMA3(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %16, 0x3f9e0652\n"

template <int KIND, int BODY16>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *cyc, float a, float b, int reps) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < (BODY16 > 1024 ? 1 : reps); it++) {
#define BODY(ops) asm volatile(".rept %18\n" ops ".endr\n" \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), \
                       "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) \
                     : "v"(a), "v"(b), "n"(BODY16))
        if constexpr (KIND == 0) BODY(OPS16(FMAC, FMAC, FMAC, FMAC));
        if constexpr (KIND == 1) BODY(OPS16(FMA3, FMA3, FMA3, FMA3));
        if constexpr (KIND == 2) BODY(OPS16(FMAAK, FMAAK, FMAAK, FMAAK));
        if constexpr (KIND == 3) BODY(OPS16(FMAC, FMA3, FMAC, FMAAK));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int BODY16>
int run(float *d_out, unsigned long long *d_cyc, int total16, const char *name, double bytes_per_inst) {
    const int reps = BODY16 > 1024 ? 1 : std::max(1, total16 / BODY16);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 1024 * wps;
        for (int r = 0; r < 3; r++) k<KIND, BODY16><<<blocks, 64>>>(d_out, d_cyc, 1.0001f, 0.5f, reps);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double ninst = 16.0 * BODY16 * reps, c = (double)h[blocks / 2] / ninst;
        printf("%-22s code %7.0f B x %4d  waves/SIMD %d  %6.2f cycles/instr/wave (median; slowest wave %6.2f)  SIMD issue interval %5.2f cycles  %5.2f B/cycle/SIMD\n",
               name, 16.0 * BODY16 * bytes_per_inst, reps, wps, c, (double)h[blocks - 1] / ninst, c / wps, bytes_per_inst * wps / c);
    }
    return 0;
}

template <int BODY16>
int all(float *d_out, unsigned long long *d_cyc, int total16) {
    if (run<0, BODY16>(d_out, d_cyc, total16, "VOP2 4 B (v_fmac)", 4)) return 1;
    if (run<1, BODY16>(d_out, d_cyc, total16, "VOP3 8 B (v_fma)", 8)) return 1;
    if (run<2, BODY16>(d_out, d_cyc, total16, "VOP2+lit 8 B (v_fmaak)", 8)) return 1;
    if (run<3, BODY16>(d_out, d_cyc, total16, "kernel mix 6 B", 6)) return 1;
    return 0;
}

int main() {
    float *d_out; unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_out, sizeof(float) * 64 * 4096)); CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 4096));
    const int total16 = 4096;     // 65536 instructions per wave
    if (all<32>(d_out, d_cyc, total16)) return 1;       // 2-4 KB loop
    if (all<512>(d_out, d_cyc, total16)) return 1;      // 32-64 KB loop (s_cbranch reaches +-128 KB)
    if (all<2048>(d_out, d_cyc, 2048)) return 1;        // 128-256 KB, executed once
    return 0;
}
