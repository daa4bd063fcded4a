// Does a wave's ds_read see the ds_write the same wave issued just before it -- to an address another LANE wrote -- without an
// s_waitcnt in between?  (DS operations of one wave are executed in order by the LDS; grid_wave_sync() used to drain lgkmcnt anyway.)
// 1024 blocks x 4 waves, 20000 rounds each: lane l writes slot l of its wave's region, reads slot (l + shift) % 64 in the very next
// instruction, checks the value.  Prints the number of mismatches (expected: 0) and the time with / without the drain.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool DRAIN>
__global__ __launch_bounds__(256) void k(unsigned *bad, int rounds) {
    __shared__ unsigned s[4][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned errors = 0;
    unsigned base = (unsigned)(size_t)(&s[wave][0]);                       // LDS byte address
    for (int r = 0; r < rounds; r++) {
        const int shift = 1 + (r % 63);
        const unsigned val = (unsigned)(r * 64 + lane) * 2654435761u + blockIdx.x;
        const unsigned waddr = base + 4u * (unsigned)(((r & 7) * 64) + lane);
        const unsigned raddr = base + 4u * (unsigned)(((r & 7) * 64) + ((lane + shift) & 63));
        unsigned got;
        if (DRAIN) asm volatile("ds_write_b32 %1, %2\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %3\n s_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(waddr), "v"(val), "v"(raddr) : "memory");
        else       asm volatile("ds_write_b32 %1, %2\n ds_read_b32 %0, %3\n s_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(waddr), "v"(val), "v"(raddr) : "memory");
        const unsigned want = (unsigned)(r * 64 + ((lane + shift) & 63)) * 2654435761u + blockIdx.x;
        errors += (got != want);
    }
    if (errors) atomicAdd(bad, errors);
}

int main() {
    unsigned *d_bad; CHECK(hipMalloc(&d_bad, 4)); CHECK(hipMemset(d_bad, 0, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int rounds = 20000;
    for (int drain = 1; drain >= 0; drain--) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            if (drain) k<true><<<1024, 256>>>(d_bad, rounds); else k<false><<<1024, 256>>>(d_bad, rounds);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned bad; CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        printf("%s: %u mismatches in %d rounds x 4096 waves, %.1f ns per write+read round\n", drain ? "write, s_waitcnt, read" : "write, read (no drain)  ", bad, rounds, ms * 1e6 / rounds);
        CHECK(hipMemset(d_bad, 0, 4));
    }
    return 0;
}
