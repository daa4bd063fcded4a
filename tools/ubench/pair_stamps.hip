// Diagnostic (GPU box): what happens to a wave of the UNSPLIT iiwa-7 forward-dynamics-gradient kernel when a second wave shares
// its SIMD?  The kernel body is the generated one (same core, same staging, __launch_bounds__(256, 2): <= 256 registers, so two
// waves fit a SIMD); an s_memtime stamp is taken at the start, after the input staging, before the first value of every
// gradient column (14 stamps through the straight-line core) and at the end.  Launched as N single-wave blocks with
// N = 256, 1024 (one wave on every SIMD), 2048 (two per SIMD) and 4096 (two rounds of two): the per-segment medians show WHICH
// part of the wave stretches when a partner is present -- the arithmetic (issue / instruction fetch) or the staging / flushes
// (LDS and the memory path).  Stamps go to a buffer of their own; the outputs are not affected.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DGRID_HEADER='"<_build>/grid_iiwa7_fp32.hip.h"' tools/ubench/pair_stamps.hip -o pair_stamps
#include GRID_HEADER
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
using namespace grid_iiwa7;
typedef float T; typedef float C;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NST = 20;      // stamps per wave: 0 start, 1 staged, 2..15 columns, 16 end, 17 realtime start, 18 realtime end, 19 XCC id / CU / SIMD

template <typename Out>
struct stamp_sink {
    Out o; unsigned long long *t;
    __device__ __forceinline__ void put(const int i, const T v){
        if (i % 7 == 0){t[i/7] = __builtin_amdgcn_s_memtime();}
        o.put(i, v);
    }
};

// sink that keeps every output alive but neither stages it in LDS nor stores it: the arithmetic alone
struct null_sink {
    __device__ __forceinline__ void put(const int i, const T v){(void)i; GRID_KEEP(v);}
};

template <int MODE>      // 0: the real kernel; 1: no output staging, no stores (arithmetic + input staging only)
__global__ __launch_bounds__(GRID_MAX_THREADS, 2)
void stamped(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const T gravity, const int NUM_TIMESTEPS, unsigned long long *stamps, const int prio) {
    extern __shared__ __align__(16) unsigned char s_grid_dyn[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (prio == 1 && (blockIdx.x & 1)){__builtin_amdgcn_s_setprio(1);}       // static priority for every other block (wave-uniform)
    const grid_tile_iter it(NUM_TIMESTEPS);
    T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*3136;
    unsigned long long c1 = 0, c2 = 0, r2 = 0;
    unsigned long long t[14];
    for (int k0 = it.k0_first; k0 < NUM_TIMESTEPS; k0 += it.k0_step){
        T s_q_qd_u[21];
        grid_load_tile<T,21>(s_q_qd_u + 0, d_q_qd_u + 0, stride_q_qd_u, k0, it, NUM_TIMESTEPS, s_wave);
        GRID_KEEP(s_q_qd_u[0]); GRID_KEEP(s_q_qd_u[20]);
        c1 = __builtin_amdgcn_s_memtime();
        const grid_in_ptrs<T> in = {s_q_qd_u, s_q_qd_u + 7, s_q_qd_u + 14, nullptr, nullptr};
        if constexpr (MODE == 0){
            stamp_sink<grid_out_staged<T,98,98,49,0,98,0>> out = {{s_wave, d_df_du, k0, it.lane, it.W, NUM_TIMESTEPS}, t};
            forward_dynamics_gradient_core<T,C>(in, out, gravity);
        } else {
            stamp_sink<null_sink> out = {{}, t};
            forward_dynamics_gradient_core<T,C>(in, out, gravity);
        }
        c2 = __builtin_amdgcn_s_memtime(); r2 = __builtin_amdgcn_s_memrealtime();
    }
    if (it.lane == 0 && it.k0_first < NUM_TIMESTEPS){
        const int w = blockIdx.x*(blockDim.x/64) + it.wave_in_block;
        unsigned long long *s = stamps + (size_t)w*NST;
        s[0] = c0; s[1] = c1;
        #pragma unroll
        for (int i = 0; i < 14; i++){s[2 + i] = t[i];}
        s[16] = c2; s[17] = r0; s[18] = r2;
        unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));          // HW_REG_XCC_ID bits 0..3
        s[19] = ((unsigned long long)xcc << 32) | hwid;
    }
}

static double med(std::vector<double> v){ std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size()/2]; }

template <int MODE>
int run(int K, int prio, float *d_in, float *d_out, unsigned long long *d_st) {
    const int n = NUM_JOINTS, blocks = std::min(K/64, 2048);
    const size_t lds = 3136*sizeof(float);
    for (int r = 0; r < 300; r++) hipLaunchKernelGGL(stamped<MODE>, dim3(blocks), dim3(64), lds, 0, d_out, d_in, 3*n, 9.81f, K, d_st, prio);   // clock ramp
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 100; r++) hipLaunchKernelGGL(stamped<MODE>, dim3(blocks), dim3(64), lds, 0, d_out, d_in, 3*n, 9.81f, K, d_st, prio);
    hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)blocks*NST);
    CHECK(hipMemcpy(st.data(), d_st, st.size()*8, hipMemcpyDeviceToHost));
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int w = 0; w < blocks; w++) { rmin = std::min(rmin, st[(size_t)w*NST+17]); rmax = std::max(rmax, st[(size_t)w*NST+18]); }
    std::vector<double> seg[16], total, start;
    for (int w = 0; w < blocks; w++) {
        const unsigned long long *s = &st[(size_t)w*NST];
        for (int i = 0; i < 16; i++) seg[i].push_back((double)(s[i+1] - s[i]));
        total.push_back((double)(s[16] - s[0])); start.push_back((double)(s[17] - rmin)*10.0);
    }
    printf("mode %d K=%-7d waves=%-5d prio=%d | %.2f us per launch | first start -> last end %.2f us | last tile of a wave: total %6.0f cyc | staging %5.0f |",
           MODE, K, blocks, prio, ms*1e3/100, (rmax - rmin)*0.01, med(total), med(seg[0]));
    for (int i = 1; i < 16; i++) printf(" %5.0f", med(seg[i]));
    printf(" | start spread median %4.0f ns max %4.0f ns", med(start), *std::max_element(start.begin(), start.end()));
    {   // distribution over waves: a co-resident pair need not share the SIMD evenly
        auto pct = [](std::vector<double> v, double p){ std::sort(v.begin(), v.end()); return v[(size_t)(p*(v.size() - 1))]; };
        std::vector<double> endt; for (int w = 0; w < blocks; w++) endt.push_back((double)(st[(size_t)w*NST+18] - rmin)*10.0);
        printf(" | total cyc p0/p10/p50/p90/p100 %.0f/%.0f/%.0f/%.0f/%.0f | prefix seg p0/p10/p50/p90/p100 %.0f/%.0f/%.0f/%.0f/%.0f | wave end (ns after first start) p10/p50/p90/p100 %.0f/%.0f/%.0f/%.0f",
               pct(total, 0), pct(total, 0.1), pct(total, 0.5), pct(total, 0.9), pct(total, 1), pct(seg[3], 0), pct(seg[3], 0.1), pct(seg[3], 0.5), pct(seg[3], 0.9), pct(seg[3], 1),
               pct(endt, 0.1), pct(endt, 0.5), pct(endt, 0.9), pct(endt, 1));
    }
    // residency: waves per SIMD (key = XCC id, SE, SH, CU, SIMD from HW_REG_HW_ID) in the LAST launch
    std::map<unsigned long long, int> per_simd, per_cu;
    for (int w = 0; w < blocks; w++) {
        const unsigned long long v = st[(size_t)w*NST+19]; const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32);
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_simd[((unsigned long long)xcc << 20) | (se << 12) | (sh << 8) | (cu << 4) | simd]++; per_cu[((unsigned long long)xcc << 20) | (se << 12) | (sh << 8) | (cu << 4)]++;
    }
    std::map<int, int> hist; for (auto &kv : per_simd) hist[kv.second]++;
    printf(" | SIMDs used %zu, CUs used %zu, waves per used SIMD:", per_simd.size(), per_cu.size());
    for (auto &kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\n");
    return 0;
}

int main() {
    const int Kmax = 262144, n = NUM_JOINTS;
    std::vector<float> x((size_t)Kmax*3*n);
    for (size_t i = 0; i < x.size(); i++) x[i] = 0.37f*(float)((i*7) % 11) - 1.3f;
    float *d_in, *d_out; unsigned long long *d_st;
    CHECK(hipMalloc(&d_in, x.size()*4)); CHECK(hipMalloc(&d_out, (size_t)Kmax*2*n*n*4)); CHECK(hipMalloc(&d_st, (size_t)2048*NST*8));
    CHECK(hipMemcpy(d_in, x.data(), x.size()*4, hipMemcpyHostToDevice));
    hipFuncAttributes a; CHECK(hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&stamped<0>)));
    printf("stamped unsplit iiwa-7 dFD kernel: %d registers, %d B scratch; segments (cycles, median over waves): staging | prefix up to column 0 | columns 0..6 of d/dq | columns 0..6 of d/dqd (the last includes the final flush)\n",
           a.numRegs, (int)a.localSizeBytes);
    for (int K : {16384, 32768, 65536, 98304, 131072, 262144}) if (run<0>(K, 0, d_in, d_out, d_st)) return 1;
    for (int K : {131072, 262144}) if (run<0>(K, 1, d_in, d_out, d_st)) return 1;
    return 0;
}
