// TEST INFRASTRUCTURE: compiles the *generated header* for the HOST (hipcc --offload-host-only) and
// runs the emitted _device / _inner functions on the CPU, one configuration at a time, so the emitted
// C++ text itself (not just the tracer IR) is checked against the oracle without a GPU.
// Never part of the product path.
#include GRID_HEADER
namespace G = GRID_NS;

template <typename C>
static void fd_grad(const float *x, float *out, int K, float g) {
    const int n = G::NUM_JOINTS;
    for (int k = 0; k < K; k++) {
        const float *r = x + (size_t)k * 3 * n;
        G::forward_dynamics_gradient_device<float, C>(out + (size_t)k * 2 * n * n, r, r + n, r + 2 * n, nullptr, g);
    }
}

extern "C" {
int hh_num_joints() { return G::NUM_JOINTS; }
void hh_fd_grad_f32(const float *x, float *out, int K, float g) { fd_grad<float>(x, out, K, g); }
void hh_fd_grad_f64(const float *x, float *out, int K, float g) { fd_grad<double>(x, out, K, g); }
void hh_id(const float *x, const float *qdd, float *out, int K, float g) {
    const int n = G::NUM_JOINTS;
    for (int k = 0; k < K; k++) {
        const float *r = x + (size_t)k * 3 * n;
        if (qdd) G::inverse_dynamics_device<float>(out + (size_t)k * n, r, r + n, qdd + (size_t)k * n, nullptr, g);
        else     G::inverse_dynamics_device<float>(out + (size_t)k * n, r, r + n, nullptr, g);
    }
}
void hh_minv(const float *x, float *out, int K) {
    const int n = G::NUM_JOINTS;
    for (int k = 0; k < K; k++) G::direct_minv_device<float>(out + (size_t)k * n * n, x + (size_t)k * 3 * n, nullptr);
}
void hh_fd(const float *x, float *out, int K, float g) {
    const int n = G::NUM_JOINTS;
    for (int k = 0; k < K; k++) { const float *r = x + (size_t)k * 3 * n; G::forward_dynamics_device<float>(out + (size_t)k * n, r, r + n, r + 2 * n, nullptr, g); }
}
// the pointer-style _inner tier chained exactly as the reference chains it inside its fused kernel
// (algorithms/_forward_dynamics_gradient.py:15-25): load_update_XImats_helpers -> direct_minv_inner ->
// inverse_dynamics_inner (qdd = 0) -> forward_dynamics_finish -> inverse_dynamics_inner_vaf -> inverse_dynamics_gradient_inner
void hh_inner_chain(const float *x, float *qdd_out, float *dc_du_out, int K, float g) {
    const int n = G::NUM_JOINTS;
    float *XI = new float[G::XIMATS_LANE_COUNT], *Minv = new float[n * n], *c = new float[n], *vaf = new float[18 * n];
    for (int k = 0; k < K; k++) {
        const float *q = x + (size_t)k * 3 * n, *qd = q + n, *u = q + 2 * n;
        float *qdd = qdd_out + (size_t)k * n;
        G::load_update_XImats_helpers<float>(XI, q, nullptr, nullptr);
        G::direct_minv_inner<float>(Minv, q, XI, nullptr);
        G::inverse_dynamics_inner<float>(c, vaf, q, qd, XI, nullptr, g);
        G::forward_dynamics_finish<float>(qdd, u, c, Minv);
        G::inverse_dynamics_inner_vaf<float>(vaf, q, qd, qdd, XI, nullptr, g);
        G::inverse_dynamics_gradient_inner<float>(dc_du_out + (size_t)k * 2 * n * n, q, qd, vaf, XI, nullptr, g);
    }
    delete[] XI; delete[] Minv; delete[] c; delete[] vaf;
}
}
extern "C" void hh_sincos(const float *x, float *s, float *c, int K) { for (int k = 0; k < K; k++) G::grid_sincos(x[k], s + k, c + k); }
// the spatial-algebra device library (public surface of the reference's header, helpers/_spatial_algebra_helpers.py:35-257):
// out = [mxK(x) | mxK_scaled(x, alpha) | y + mxK_peq(x) | y + mxK_peq_scaled(x, alpha)] for K = 0..5 through the runtime-selected
// mxX family (6 x 4 x 6 values), then fx(x) (36, column-major), fx_zeroed on zeros (36), fx_times_v(x, y) (6), y + fx_times_v_peq(x, y) (6),
// dot_prod<6,1,1>(x, y) and dot_prod<3,2,1>(x, y) (2)
extern "C" void hh_spatial(const float *x, const float *y, float alpha, float *out) {
    float *o = out;
    for (int k = 0; k < 6; k++) {
        G::mxX<float>(o, x, k); o += 6;
        G::mxX_scaled<float>(o, x, alpha, k); o += 6;
        for (int i = 0; i < 6; i++) o[i] = y[i];
        G::mxX_peq<float>(o, x, k); o += 6;
        for (int i = 0; i < 6; i++) o[i] = y[i];
        G::mxX_peq_scaled<float>(o, x, alpha, k); o += 6;
    }
    G::fx<float>(o, x); o += 36;
    for (int i = 0; i < 36; i++) o[i] = 0.0f;
    G::fx_zeroed<float>(o, x); o += 36;
    G::fx_times_v<float>(o, x, y); o += 6;
    for (int i = 0; i < 6; i++) o[i] = y[i];
    G::fx_times_v_peq<float>(o, x, y); o += 6;
    float xm[6]; for (int i = 0; i < 6; i++) xm[i] = x[i];
    o[0] = G::dot_prod<float, 6, 1, 1>(x, y); o[1] = G::dot_prod<float, 3, 2, 1>(xm, y);
}
// launch-shape sanitiser of the dim3 host wrappers: grid_launch_dims (illegal shapes -> the suggested one) followed by
// grid_fold_launch_z (z extents folded into y: the kernels number threads and blocks by x and y only).  io = {bx, by, bz, tx, ty, tz}
extern "C" void hh_launch_shape(int *io, int num_timesteps) {
    dim3 blocks, threads;
    G::grid_launch_dims(dim3(io[0], io[1], io[2]), dim3(io[3], io[4], io[5]), num_timesteps, &blocks, &threads);
    G::grid_fold_launch_z(&blocks, &threads);
    io[0] = blocks.x; io[1] = blocks.y; io[2] = blocks.z; io[3] = threads.x; io[4] = threads.y; io[5] = threads.z;
}
