// Micro-benchmark: steady-state cost of writing the Atlas-30 gradient output (K x 1800 floats, 472 MB at K = 65536) with
// different store shapes, one wave per 64 rows, 1024 single-wave blocks:
//   RUN = 30 : per column, lane group g of 32 lanes writes 30 contiguous floats of row 2t+g (what grid_out_staged does)
//   RUN = 60 / 120 / 1800: the same with longer contiguous runs per row (1800 = whole row, 64 lanes sweep it)
// Each kernel is launched 6 times; the first (cold caches / clean MALL) and the average of the rest are reported.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ROW = 1800;

template <int RUN, bool NT>
__global__ __launch_bounds__(64) void wr(float *out, int K, float v) {
    const int lane = threadIdx.x;
    for (int k0 = blockIdx.x * 64; k0 < K; k0 += gridDim.x * 64) {
        float *base = out + (size_t)k0 * ROW;
        if (RUN >= 64) {
            // 64 lanes sweep RUN contiguous floats of one row at a time
            for (int c = 0; c < ROW / RUN; c++)
                for (int r = 0; r < 64; r++)
                    for (int i = lane; i < RUN; i += 64) { if (NT) __builtin_nontemporal_store(v + i, &base[(size_t)r * ROW + c * RUN + i]); else base[(size_t)r * ROW + c * RUN + i] = v + i; }
        } else {
            constexpr int P = RUN <= 32 ? 32 : 64, G = 64 / P;
            const int g = lane / P, ii = lane % P;
            for (int c = 0; c < ROW / RUN; c++) {
                if (ii < RUN) {
#pragma unroll 8
                    for (int t = 0; t < 64 / G; t++) { if (NT) __builtin_nontemporal_store(v + t, &base[(size_t)(t * G + g) * ROW + c * RUN + ii]); else base[(size_t)(t * G + g) * ROW + c * RUN + ii] = v + t; }
                }
            }
        }
    }
}

template <int RUN, bool NT>
int run(float *d_out, int K) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float first = 0, rest = 0;
    for (int rep = 0; rep < 6; rep++) {
        CHECK(hipEventRecord(e0));
        wr<RUN, NT><<<1024, 64>>>(d_out, K, 1.0f + rep);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) first = ms; else rest += ms / 5;
    }
    const double bytes = (double)K * ROW * 4;
    printf("%s run %4d floats  K %6d  first %7.1f us  steady %7.1f us  = %.2f TB/s\n", NT ? "nt   " : "plain", RUN, K, first * 1e3, rest * 1e3, bytes / (rest * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int K = 65536;
    float *d_out; CHECK(hipMalloc(&d_out, sizeof(float) * (size_t)K * ROW));
    for (int K2 : {32768, 65536}) {
        if (run<30, false>(d_out, K2)) return 1;
        if (run<30, true>(d_out, K2)) return 1;
        if (run<60, false>(d_out, K2)) return 1;
        if (run<60, true>(d_out, K2)) return 1;
        if (run<120, false>(d_out, K2)) return 1;
        if (run<1800, false>(d_out, K2)) return 1;
        if (run<1800, true>(d_out, K2)) return 1;
    }
    return 0;
}
