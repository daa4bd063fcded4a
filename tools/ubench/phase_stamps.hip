// Diagnostic (GPU box): where does a wave of the headline kernel spend its time?  Re-runs the body of
// forward_dynamics_gradient_kernel_split4 (iiwa-7, generated header) with s_memtime / s_memrealtime stamps around the input
// staging and around the core (+ output flushes).  Stamps go to a buffer of their own; the outputs are not affected.
// built by __graft_entry__.build_ubench() (which also writes phase_stamps_cases.inc from the generated header)
#include GRID_HEADER
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace grid_iiwa7;
typedef float T; typedef float C;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(GRID_MAX_THREADS)
void stamped(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const T gravity, const int NUM_TIMESTEPS, unsigned long long *stamps) {
    extern __shared__ __align__(16) unsigned char s_grid_dyn[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const grid_tile_iter it(NUM_TIMESTEPS, 4);
    T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*3136;
    unsigned long long c1 = 0, c2 = 0, r1 = 0, r2 = 0;
    for (int k0 = it.k0_first; k0 < NUM_TIMESTEPS; k0 += it.k0_step){
        T s_q_qd_u[21];
        grid_load_tile<T,21>(s_q_qd_u + 0, d_q_qd_u + 0, stride_q_qd_u, k0, it, NUM_TIMESTEPS, s_wave);
        GRID_KEEP(s_q_qd_u[0]); GRID_KEEP(s_q_qd_u[20]);
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        const grid_in_ptrs<T> in = {s_q_qd_u, s_q_qd_u + 7, s_q_qd_u + 14, nullptr, nullptr};
        // the `switch (it.part){...}` of forward_dynamics_gradient_kernel_split4, copied from the generated header by
        // __graft_entry__.build_ubench() (column sets and their sinks are the generator's choice)
#include "phase_stamps_cases.inc"
        c2 = __builtin_amdgcn_s_memtime(); r2 = __builtin_amdgcn_s_memrealtime();
    }
    if (it.lane == 0 && it.k0_first < NUM_TIMESTEPS){
        const int w = blockIdx.x*(blockDim.x/64) + it.wave_in_block;
        unsigned long long *s = stamps + (size_t)w*8;
        s[0] = c0; s[1] = c1; s[2] = c2; s[3] = r0; s[4] = r1; s[5] = r2; s[6] = it.part;
    }
}

int main() {
    const int K = 16384, n = NUM_JOINTS, tiles = K/64, blocks = tiles;       // one block of 4 waves per tile: its four column groups
    std::vector<float> x((size_t)K*3*n);
    for (size_t i = 0; i < x.size(); i++) x[i] = 0.37f*(float)((i*7) % 11) - 1.3f;
    float *d_in, *d_out; unsigned long long *d_st;
    CHECK(hipMalloc(&d_in, x.size()*4)); CHECK(hipMalloc(&d_out, (size_t)K*2*n*n*4)); CHECK(hipMalloc(&d_st, (size_t)blocks*4*8*8));
    CHECK(hipMemcpy(d_in, x.data(), x.size()*4, hipMemcpyHostToDevice));
    const size_t lds = 4*3136*sizeof(float);
    for (int r = 0; r < 2000; r++) hipLaunchKernelGGL(stamped, dim3(blocks), dim3(256), lds, 0, d_out, d_in, 3*n, 9.81f, K, d_st);   // clock ramp
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 200; r++) hipLaunchKernelGGL(stamped, dim3(blocks), dim3(256), lds, 0, d_out, d_in, 3*n, 9.81f, K, d_st);
    hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int waves = blocks*4;
    std::vector<unsigned long long> st((size_t)waves*8);
    CHECK(hipMemcpy(st.data(), d_st, st.size()*8, hipMemcpyDeviceToHost));
    printf("kernel (with stamps) %.2f us per launch\n", ms*1e3/200);
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int w = 0; w < waves; w++) { rmin = std::min(rmin, st[(size_t)w*8+3]); rmax = std::max(rmax, st[(size_t)w*8+5]); }
    printf("first wave start -> last wave end: %.2f us (s_memrealtime, 100 MHz)\n", (rmax - rmin)*0.01);
    for (int part = 0; part < 4; part++) {
        std::vector<double> load_c, core_c, load_ns, core_ns, start_ns;
        for (int w = 0; w < waves; w++) { const unsigned long long *s = &st[(size_t)w*8]; if ((int)s[6] != part) continue;
            load_c.push_back((double)(s[1]-s[0])); core_c.push_back((double)(s[2]-s[1])); load_ns.push_back((s[4]-s[3])*10.0); core_ns.push_back((s[5]-s[4])*10.0);
            start_ns.push_back((s[3]-rmin)*10.0); }
        auto med = [](std::vector<double> v){ std::sort(v.begin(), v.end()); return v[v.size()/2]; };
        auto mx = [](std::vector<double> v){ return *std::max_element(v.begin(), v.end()); };
        printf("part %d: staging %7.0f cyc (%5.0f ns)  core+flush %7.0f cyc (%5.0f ns)  clock %.2f GHz | wave start after first: median %5.0f ns, max %5.0f ns\n",
               part, med(load_c), med(load_ns), med(core_c), med(core_ns), med(core_c)/med(core_ns), med(start_ns), mx(start_ns));
    }
    return 0;
}
