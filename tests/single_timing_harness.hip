// TEST INFRASTRUCTURE (GPU): a GRiD-style main() over the generated header.  For every algorithm it runs the reference-style
// host wrapper (mode 0) on one configuration and the *_single_timing twin (reference mode 1: the same configuration evaluated
// `reps` times inside one kernel) and checks that both leave the same result in the host buffers.
// Large robots: built with -DGRID_EXTERN_KERNELS and linked against the robot's kernel library (only the latency twins are compiled
// here), and with -DGRID_ST_REL_TOL=<tol>: their mode-0 forward-dynamics gradient is served by the tile-cooperative kernel, whose
// arithmetic differs from the twin's lane-per-configuration core in rounding -- the comparison is then norm-wise instead of bitwise.
#include GRID_HEADER
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace GRID_NS;
#ifdef GRID_EXTERN_KERNELS
typedef float T;
GRID_FOR_EACH_KERNEL_INST(extern template)
#endif

static int compare(const char *what, const float *a, const float *b, int count) {
#ifdef GRID_ST_REL_TOL
    float scale = 0.f, worst = 0.f;
    for (int i = 0; i < count; i++) { scale = std::fmax(scale, std::fabs(a[i])); worst = std::fmax(worst, std::fabs(a[i] - b[i])); }
    if (!(worst <= (float)(GRID_ST_REL_TOL) * scale)) { printf("MISMATCH %s: max |diff| %g against scale %g\n", what, worst, scale); return 1; }
    printf("MATCH %s %d (max |diff| / scale %.2e)\n", what, count, scale > 0.f ? worst / scale : 0.f);
    return 0;
#else
    for (int i = 0; i < count; i++) {
        if (std::memcmp(a + i, b + i, sizeof(float)) != 0) { printf("MISMATCH %s [%d] %g vs %g\n", what, i, a[i], b[i]); return 1; }
    }
    printf("MATCH %s %d\n", what, count);
    return 0;
#endif
}

int main() {
    const int n = NUM_JOINTS, reps = 5;
    hipStream_t *streams = init_grid<float>();
    robotModel<float> *d_robotModel = init_robotModel<float>();
    gridData<float> *hd = init_gridData<float>(1);
    for (int i = 0; i < 3 * n; i++) { hd->h_q_qd_u[i] = 0.37f * (float)((i * 7) % 11) - 1.3f; }
    for (int i = 0; i < 2 * n; i++) { hd->h_q_qd[i] = hd->h_q_qd_u[i]; }
    for (int i = 0; i < n; i++) { hd->h_q[i] = hd->h_q_qd_u[i]; }
    const dim3 blocks(1, 1, 1), threads(SUGGESTED_THREADS, 1, 1);
    const float g = 9.81f;
    int bad = 0;
    std::vector<float> keep(2 * n * n);

    inverse_dynamics<float>(hd, d_robotModel, g, 1, blocks, threads, streams);
    std::memcpy(keep.data(), hd->h_c, n * sizeof(float)); std::memset(hd->h_c, 0, n * sizeof(float));
    inverse_dynamics_single_timing<float>(hd, d_robotModel, g, reps, blocks, threads, streams);
    bad += compare("ID", keep.data(), hd->h_c, n);

    direct_minv<float>(hd, d_robotModel, 1, blocks, threads, streams);
    std::memcpy(keep.data(), hd->h_Minv, n * n * sizeof(float)); std::memset(hd->h_Minv, 0, n * n * sizeof(float));
    direct_minv_single_timing<float>(hd, d_robotModel, reps, blocks, threads, streams);
    bad += compare("Minv", keep.data(), hd->h_Minv, n * n);

    forward_dynamics<float>(hd, d_robotModel, g, 1, blocks, threads, streams);
    std::memcpy(keep.data(), hd->h_qdd, n * sizeof(float));
    std::vector<float> qdd(hd->h_qdd, hd->h_qdd + n);
    std::memset(hd->h_qdd, 0, n * sizeof(float));
    forward_dynamics_single_timing<float>(hd, d_robotModel, g, reps, blocks, threads, streams);
    bad += compare("FD", keep.data(), hd->h_qdd, n);

    inverse_dynamics_gradient<float>(hd, d_robotModel, g, 1, blocks, threads, streams);
    std::memcpy(keep.data(), hd->h_dc_du, 2 * n * n * sizeof(float)); std::memset(hd->h_dc_du, 0, 2 * n * n * sizeof(float));
    inverse_dynamics_gradient_single_timing<float>(hd, d_robotModel, g, reps, blocks, threads, streams);
    bad += compare("ID_DU", keep.data(), hd->h_dc_du, 2 * n * n);

    // USE_QDD_FLAG variant (qdd = the forward-dynamics result)
    std::memcpy(hd->h_qdd, qdd.data(), n * sizeof(float));
    inverse_dynamics_gradient<float, true>(hd, d_robotModel, g, 1, blocks, threads, streams);
    std::memcpy(keep.data(), hd->h_dc_du, 2 * n * n * sizeof(float)); std::memset(hd->h_dc_du, 0, 2 * n * n * sizeof(float));
    inverse_dynamics_gradient_single_timing<float, true>(hd, d_robotModel, g, reps, blocks, threads, streams);
    bad += compare("ID_DU(qdd)", keep.data(), hd->h_dc_du, 2 * n * n);

    forward_dynamics_gradient<float>(hd, d_robotModel, g, 1, blocks, threads, streams);
    std::memcpy(keep.data(), hd->h_df_du, 2 * n * n * sizeof(float)); std::memset(hd->h_df_du, 0, 2 * n * n * sizeof(float));
    forward_dynamics_gradient_single_timing<float>(hd, d_robotModel, g, reps, blocks, threads, streams);
    bad += compare("FD_DU", keep.data(), hd->h_df_du, 2 * n * n);

    close_grid<float>(streams, d_robotModel, hd);
    printf(bad ? "FAILED %d\n" : "ALL MATCH\n", bad);
    return bad;
}
