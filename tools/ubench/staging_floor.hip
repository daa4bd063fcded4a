// Micro-benchmark: the fixed cost of a lane-per-configuration kernel (launch + staging in/out) without any dynamics.
#include GRID_HEADER
#include <cstdio>
namespace G = GRID_NS;
typedef float T;
template <int MODE>
__global__ __launch_bounds__(256) void floor_kernel(T *d_out, const T *d_in, const int stride, const int K) {
    extern __shared__ __align__(16) unsigned char s_grid_dyn[];
    const G::grid_tile_iter it(K);
    T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block * 64 * 21;
    for (int k0 = it.k0_first; k0 < K; k0 += it.k0_step) {
        T s_in[21];
        G::grid_load_tile<T, 21>(s_in, d_in, stride, k0, it, K, s_wave);
        if (MODE == 0) {            // 7 outputs (like RNEA / FD)
            G::grid_out_staged<T, 7, 7, 7, 0, 7, 0> out = {s_wave, d_out, k0, it.lane, it.W, K};
            for (int i = 0; i < 7; i++) out.put(i, s_in[i] + s_in[i + 7] * s_in[i + 14]);
        } else if (MODE == 1) {     // 7 outputs + 6 sincos
            G::grid_out_staged<T, 7, 7, 7, 0, 7, 0> out = {s_wave, d_out, k0, it.lane, it.W, K};
            float acc = 0;
            for (int i = 0; i < 7; i++) { float s, c; G::grid_sincos(s_in[i], &s, &c); out.put(i, s * s_in[i + 7] + c * s_in[i + 14]); }
        } else if (MODE == 2) {     // 98 outputs in 2 chunks (like FD gradient)
            G::grid_out_staged<T, 98, 98, 49, 0, 98, 0> out = {s_wave, d_out, k0, it.lane, it.W, K};
            for (int i = 0; i < 98; i++) out.put(i, s_in[i % 21] * 1.5f);
        } else if (MODE == 3) {     // 98 outputs, direct strided stores (no staging)
            for (int i = 0; i < 98; i++) if (k0 + it.lane < K) d_out[(size_t)(k0 + it.lane) * 98 + i] = s_in[i % 21] * 1.5f;
        } else if (MODE == 4) {     // 98 outputs, one flat chunk through LDS, unpredicated flat dword copy
            T *w = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block * 64 * 98;
            for (int i = 0; i < 98; i++) w[it.lane * 98 + i] = s_in[i % 21] * 1.5f;
            G::grid_wave_sync();
            T *dst = d_out + (size_t)k0 * 98;
            #pragma unroll
            for (int t = 0; t < 98; t++) dst[t * 64 + it.lane] = w[t * 64 + it.lane];
        } else if (MODE == 5) {     // 98 outputs, one flat chunk through LDS, dwordx4 copy
            T *w = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block * 64 * 98;
            for (int i = 0; i < 98; i++) w[it.lane * 98 + i] = s_in[i % 21] * 1.5f;
            G::grid_wave_sync();
            float4 *dst = reinterpret_cast<float4 *>(d_out + (size_t)k0 * 98);
            const float4 *src = reinterpret_cast<const float4 *>(w);
            #pragma unroll
            for (int t = 0; t < 25; t++) { const int f = t * 64 + it.lane; if (f < 64 * 98 / 4) dst[f] = src[f]; }
        } else if (MODE == 6) {     // 2 chunks of 49 like MODE 2 but unpredicated (full tiles)
            for (int c = 0; c < 2; c++) {
                for (int i = 0; i < 49; i++) s_wave[it.lane * 49 + i] = s_in[i % 21] * 1.5f;
                G::grid_wave_sync();
                T *dst = d_out + (size_t)k0 * 98 + c * 49;
                #pragma unroll
                for (int t = 0; t < 49; t++) { const int f = t * 64 + it.lane; const int cfg = f / 49; const int i = f - cfg * 49; dst[(size_t)cfg * 98 + i] = s_wave[f]; }
                G::grid_wave_sync();
            }
        }
    }
}
__global__ void empty_kernel(T *d_out) { if (threadIdx.x == 1000) d_out[0] = 1; }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename F> float time_us(F launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e0); for (int i = 0; i < reps; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps;
}
int main() {
    const int K = 16384;
    T *d_in, *d_out; CHECK(hipMalloc(&d_in, sizeof(T) * K * 21)); CHECK(hipMalloc(&d_out, sizeof(T) * K * 98));
    CHECK(hipMemset(d_in, 0, sizeof(T) * K * 21));
    const size_t lds = 64 * 21 * 4 > 64 * 49 * 4 ? 64 * 21 * 4 : 64 * 49 * 4;
    printf("empty kernel (256x64):            %6.2f us\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(64), 0, 0, d_out); }, 500));
    printf("stage in 21 / out 7:              %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<0>, dim3(256), dim3(64), lds, 0, d_out, d_in, 21, K); }, 500));
    printf("stage in 21 / 6 sincos / out 7:   %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<1>, dim3(256), dim3(64), lds, 0, d_out, d_in, 21, K); }, 500));
    printf("stage in 21 / out 98 (2 chunks):  %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<2>, dim3(256), dim3(64), lds, 0, d_out, d_in, 21, K); }, 500));
    printf("out 98 direct strided stores:     %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<3>, dim3(256), dim3(64), lds, 0, d_out, d_in, 21, K); }, 500));
    printf("out 98 flat dword copy via LDS:   %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<4>, dim3(256), dim3(64), 64 * 98 * 4, 0, d_out, d_in, 21, K); }, 500));
    printf("out 98 flat dwordx4 copy via LDS: %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<5>, dim3(256), dim3(64), 64 * 98 * 4, 0, d_out, d_in, 21, K); }, 500));
    printf("out 98 2 chunks unpredicated:     %6.2f us\n", time_us([&] { hipLaunchKernelGGL(floor_kernel<6>, dim3(256), dim3(64), lds, 0, d_out, d_in, 21, K); }, 500));
    CHECK(hipDeviceSynchronize());
    return 0;
}
