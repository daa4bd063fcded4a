// Micro-benchmark: the Atlas-30 gradient output (K rows of 1800 floats, 7200 B: 472 MB at K = 65536) written as the lean kernel's
// flushes write it -- one wave per tile of 64 rows, per flush a piece of every row -- with pieces that ignore or respect the 32-byte
// sectors of the memory side (a row starts on a sector boundary: 7200 = 225 x 32):
//   run30   : 30-float pieces at column offsets (120 B, 8-byte aligned), dwordx2, 15 lanes per row, 4 rows per instruction (shipped)
//   piece32 : 32-float pieces at multiples of 128 B, dwordx2, 16 lanes per row, 4 rows per instruction (+ one 8-float tail piece)
//   piece32q: the same with dwordx4, 8 lanes per row, 8 rows per instruction
//   piece64q: 64-float pieces, dwordx4, 16 lanes per row, 4 rows per instruction
// Each kernel is launched 6 times; the first and the average of the rest are reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ROW = 1800;
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// MODE 0: run30 (dwordx2); 1: piece32 (dwordx2); 2: piece32q (dwordx4); 3: piece64q (dwordx4)
template <int MODE>
__global__ __launch_bounds__(64) void wr(float *out, int K, float v) {
    const int lane = threadIdx.x;
    for (int k0 = blockIdx.x * 64; k0 < K; k0 += gridDim.x * 64) {
        float *base = out + (size_t)k0 * ROW;
        if (MODE == 0) {
            const int g = lane / 16, ii = min(lane % 16, 14);
            for (int c = 0; c < ROW / 30; c++) {
#pragma unroll 8
                for (int t = 0; t < 16; t++) { float2v x = {v + t, v + c}; *(float2v *)&base[(size_t)(t * 4 + g) * ROW + c * 30 + 2 * ii] = x; }
            }
        } else if (MODE == 1) {
            const int g = lane / 16, ii = lane % 16;
            for (int c = 0; c < ROW / 32; c++) {
#pragma unroll 8
                for (int t = 0; t < 16; t++) { float2v x = {v + t, v + c}; *(float2v *)&base[(size_t)(t * 4 + g) * ROW + c * 32 + 2 * ii] = x; }
            }
            { const int g4 = lane / 4, i4 = lane % 4;      // tail: 8 floats = 4 pairs per row, 16 rows per instruction
              for (int t = 0; t < 4; t++) { float2v x = {v + t, v}; *(float2v *)&base[(size_t)(t * 16 + g4) * ROW + 1792 + 2 * i4] = x; } }
        } else if (MODE == 2) {
            const int g = lane / 8, ii = lane % 8;
            for (int c = 0; c < ROW / 32; c++) {
#pragma unroll 8
                for (int t = 0; t < 8; t++) { float4v x = {v + t, v + c, v, v}; *(float4v *)&base[(size_t)(t * 8 + g) * ROW + c * 32 + 4 * ii] = x; }
            }
            { const int g2 = lane / 2, i2 = lane % 2;
              for (int t = 0; t < 2; t++) { float4v x = {v + t, v, v, v}; *(float4v *)&base[(size_t)(t * 32 + g2) * ROW + 1792 + 4 * i2] = x; } }
        } else {
            const int g = lane / 16, ii = lane % 16;
            for (int c = 0; c < ROW / 64; c++) {
#pragma unroll 8
                for (int t = 0; t < 16; t++) { float4v x = {v + t, v + c, v, v}; *(float4v *)&base[(size_t)(t * 4 + g) * ROW + c * 64 + 4 * ii] = x; }
            }
            { const int g2 = lane / 2, i2 = lane % 2;       // tail: 1792 .. 1800
              for (int t = 0; t < 2; t++) { float4v x = {v + t, v, v, v}; *(float4v *)&base[(size_t)(t * 32 + g2) * ROW + 1792 + 4 * i2] = x; } }
        }
    }
}

template <int MODE>
int run(float *d_out, int K, const char *name) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float first = 0, rest = 0;
    for (int rep = 0; rep < 6; rep++) {
        CHECK(hipEventRecord(e0));
        wr<MODE><<<1024, 64>>>(d_out, K, 1.0f + rep);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) first = ms; else rest += ms / 5;
    }
    const double bytes = (double)K * ROW * 4;
    printf("%-9s K %6d  first %7.1f us  steady %7.1f us  = %.2f TB/s\n", name, K, first * 1e3, rest * 1e3, bytes / (rest * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char **argv) {
    const int K = 131072;
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    float *d_out; CHECK(hipMalloc(&d_out, sizeof(float) * (size_t)K * ROW));
    for (int K2 : {16384, 65536, 131072}) {
        if ((only < 0 || only == 0) && run<0>(d_out, K2, "run30")) return 1;
        if ((only < 0 || only == 1) && run<1>(d_out, K2, "piece32")) return 1;
        if ((only < 0 || only == 2) && run<2>(d_out, K2, "piece32q")) return 1;
        if ((only < 0 || only == 3) && run<3>(d_out, K2, "piece64q")) return 1;
    }
    return 0;
}
