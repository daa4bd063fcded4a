// Micro-benchmark: time per launch of N dependent kernels issued (a) one by one into a stream, (b) as one captured hipGraph of N
// kernel nodes -- for an empty kernel and for a kernel that keeps 1024 single-wave blocks busy for ~7 us (the shape of the headline
// launch).  What a graph saves per dependent launch is the answer to "capture launch-bound inner loops in hipGraphs".
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void empty_kernel(float *p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.0f; }

__global__ __launch_bounds__(64) void busy_kernel(float *p, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) { a = fmaf(a, b, 1e-7f); b = fmaf(b, 0.99999f, 1e-9f); }
    if (a == 123.456f) p[blockIdx.x] = a + b;       // (never true: keeps the loop alive)
}

template <typename F>
static int measure(const char *name, int blocks, F launch) {
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int N = 200;
    for (int i = 0; i < 2000; i++) launch(s);                  // ramp the clocks
    CHECK(hipStreamSynchronize(s));
    float best_stream = 1e9f, best_graph = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < N; i++) launch(s);
        CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_stream) best_stream = ms;
    }
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; i++) launch(s);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, s));
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_graph) best_graph = ms;
    }
    printf("%-28s blocks %5d | stream %6.2f us per launch | graph of %d nodes %6.2f us per node\n", name, blocks, best_stream * 1e3 / N, N, best_graph * 1e3 / N);
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipStreamDestroy(s));
    return 0;
}

int main() {
    float *d; CHECK(hipMalloc(&d, 1 << 20));
    if (measure("empty kernel", 1, [&](hipStream_t s) { empty_kernel<<<1, 64, 0, s>>>(d); })) return 1;
    if (measure("empty kernel", 1024, [&](hipStream_t s) { empty_kernel<<<1024, 64, 0, s>>>(d); })) return 1;
    for (int iters : {1000, 2000, 4000}) {
        char name[64]; snprintf(name, sizeof name, "busy kernel (%d iterations)", iters);
        if (measure(name, 1024, [&](hipStream_t s) { busy_kernel<<<1024, 64, 0, s>>>(d, iters); })) return 1;
    }
    return 0;
}
