// TEST INFRASTRUCTURE (GPU): executes the parts of the generated header's public surface that the C ABI does not reach --
// every *_compute_only and *_launch host wrapper, every USE_COMPRESSED_MEM=true instantiation, inverse_dynamics_vaf_device,
// the _device tier called from a user kernel, and a user kernel that runs the pointer-style _inner chain per lane
// (load_update_XImats_helpers -> direct_minv_inner -> inverse_dynamics_inner -> forward_dynamics_finish ->
// inverse_dynamics_inner_vaf -> inverse_dynamics_gradient_inner), exactly as a GRiD user would write them
// (reference: algorithms/_inverse_dynamics.py:311-352,423-495, _forward_dynamics_gradient.py:59-99; banner
// GRiDCodeGenerator.py:243-279).  Built as a shared library by gridcodegenerator_amd.host.build_api_harness() and driven
// from tests/test_api_surface.py through ctypes.  Never part of the product path.
//
//   hipcc --offload-arch=gfx950 -O1 -ffp-contract=off -fPIC -shared -DGRID_HEADER='"..."' -DGRID_NS=grid_<robot>
//         [-DGRID_EXTERN_KERNELS -L<build> -lgrid_<robot>_<precision>]  tests/api_surface_harness.hip
#define GRID_ERRORS_RETURN 1
#include GRID_HEADER
#include <cstring>
#include <string>
namespace G = GRID_NS;
typedef float T;
#ifdef GRID_EXTERN_KERNELS
GRID_FOR_EACH_KERNEL_INST(extern template)     // the library's kernel objects are linked in (large robots: minutes each)
#endif

static const int N = G::NUM_JOINTS;
static std::string g_err;

// ---- user kernels over the _device and _inner tiers: one lane per configuration, lane-private arrays -------------------
__global__ void vaf_device_kernel(T *d_vaf, const T *d_q_qd_u, const T *d_qdd, const G::robotModel<T> *d_robotModel, T gravity, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    T s_q[N], s_qd[N], s_qdd[N], s_vaf[18 * N];
    for (int i = 0; i < N; i++) { s_q[i] = d_q_qd_u[(size_t)k * 3 * N + i]; s_qd[i] = d_q_qd_u[(size_t)k * 3 * N + N + i]; }
    if (d_qdd != nullptr) {
        for (int i = 0; i < N; i++) s_qdd[i] = d_qdd[(size_t)k * N + i];
        G::inverse_dynamics_vaf_device<T>(s_vaf, s_q, s_qd, s_qdd, d_robotModel, gravity);
    } else {
        G::inverse_dynamics_vaf_device<T>(s_vaf, s_q, s_qd, d_robotModel, gravity);
    }
    for (int i = 0; i < 18 * N; i++) d_vaf[(size_t)k * 18 * N + i] = s_vaf[i];
}

// out row: [c (n) | Minv (n*n) | qdd (n) | dc_du at qdd (2*n*n)]
__global__ void device_tier_kernel(T *d_out, const T *d_q_qd_u, const G::robotModel<T> *d_robotModel, T gravity, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    T s_q[N], s_qd[N], s_u[N], s_c[N], s_qdd[N];
    for (int i = 0; i < N; i++) {
        s_q[i] = d_q_qd_u[(size_t)k * 3 * N + i]; s_qd[i] = d_q_qd_u[(size_t)k * 3 * N + N + i]; s_u[i] = d_q_qd_u[(size_t)k * 3 * N + 2 * N + i];
    }
    T *row = d_out + (size_t)k * (2 * N + 3 * N * N);
    G::inverse_dynamics_device<T>(s_c, s_q, s_qd, d_robotModel, gravity);
    for (int i = 0; i < N; i++) row[i] = s_c[i];
    G::direct_minv_device<T>(row + N, s_q, d_robotModel);               // (lane-private in the API; a global row works too)
    G::forward_dynamics_device<T>(s_qdd, s_q, s_qd, s_u, d_robotModel, gravity);
    for (int i = 0; i < N; i++) row[N + N * N + i] = s_qdd[i];
    // (large robots: the gradient core is the same one the kernels run and takes minutes to compile a second time here;
    //  their column-serial gradient is covered through inverse_dynamics_gradient_inner below)
    if constexpr (N <= 12) G::inverse_dynamics_gradient_device<T>(row + 2 * N + N * N, s_q, s_qd, s_qdd, d_robotModel, gravity);
}

// out row: [qdd (n) | Minv (n*n) | vaf at qdd (18n) | dc_du at qdd (2*n*n)]
__global__ void inner_chain_kernel(T *d_out, const T *d_q_qd_u, const G::robotModel<T> *d_robotModel, T gravity, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    T s_q[N], s_qd[N], s_u[N], s_c[N], s_qdd[N], s_XImats[G::XIMATS_LANE_COUNT];
    for (int i = 0; i < N; i++) {
        s_q[i] = d_q_qd_u[(size_t)k * 3 * N + i]; s_qd[i] = d_q_qd_u[(size_t)k * 3 * N + N + i]; s_u[i] = d_q_qd_u[(size_t)k * 3 * N + 2 * N + i];
    }
    T *row = d_out + (size_t)k * (N + N * N + 18 * N + 2 * N * N);
    T *s_Minv = row + N, *s_vaf = row + N + N * N, *s_dc_du = row + N + N * N + 18 * N;
    G::load_update_XImats_helpers<T>(s_XImats, s_q, d_robotModel, nullptr);
    G::direct_minv_inner<T>(s_Minv, s_q, s_XImats, nullptr);
    G::inverse_dynamics_inner<T>(s_c, s_vaf, s_q, s_qd, s_XImats, nullptr, gravity);
    G::forward_dynamics_finish<T>(s_qdd, s_u, s_c, s_Minv);
    for (int i = 0; i < N; i++) row[i] = s_qdd[i];
    G::inverse_dynamics_inner_vaf<T>(s_vaf, s_q, s_qd, s_qdd, s_XImats, nullptr, gravity);
    G::inverse_dynamics_gradient_inner<T>(s_dc_du, s_q, s_qd, s_vaf, s_XImats, nullptr, gravity);
}

// The spatial-algebra device library (dot_prod, mxX / mxK families, fx, fx_zeroed, fx_times_v[_peq]: reference
// helpers/_spatial_algebra_helpers.py:35-257) called from a user kernel, one lane per (x, y) pair.  out row (SPATIAL_ROW values):
// for K = 0..5: mxX | mxX_scaled | y + mxX_peq | y + mxX_peq_scaled (6 each); fx (36, column-major); fx_zeroed on zeros (36);
// fx_times_v (6); y + fx_times_v_peq (6); dot_prod<6,1,1>(x, y); dot_prod<3,2,1>(x, y); then mx0..mx5 called by NAME (6 x 6)
static const int SPATIAL_ROW = 6 * 4 * 6 + 36 + 36 + 6 + 6 + 2 + 36;
__global__ void spatial_kernel(T *d_out, const T *d_xy, T alpha, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    T x[6], y[6], o[SPATIAL_ROW];
    for (int i = 0; i < 6; i++) { x[i] = d_xy[(size_t)k * 12 + i]; y[i] = d_xy[(size_t)k * 12 + 6 + i]; }
    int at = 0;
    for (int c = 0; c < 6; c++) {
        G::mxX<T>(o + at, x, c); at += 6;
        G::mxX_scaled<T>(o + at, x, alpha, c); at += 6;
        for (int i = 0; i < 6; i++) o[at + i] = y[i];
        G::mxX_peq<T>(o + at, x, c); at += 6;
        for (int i = 0; i < 6; i++) o[at + i] = y[i];
        G::mxX_peq_scaled<T>(o + at, x, alpha, c); at += 6;
    }
    G::fx<T>(o + at, x); at += 36;
    for (int i = 0; i < 36; i++) o[at + i] = static_cast<T>(0);
    G::fx_zeroed<T>(o + at, x); at += 36;
    G::fx_times_v<T>(o + at, x, y); at += 6;
    for (int i = 0; i < 6; i++) o[at + i] = y[i];
    G::fx_times_v_peq<T>(o + at, x, y); at += 6;
    o[at] = G::dot_prod<T, 6, 1, 1>(x, y); o[at + 1] = G::dot_prod<T, 3, 2, 1>(x, y); at += 2;
    G::mx0<T>(o + at, x); G::mx1<T>(o + at + 6, x); G::mx2<T>(o + at + 12, x); G::mx3<T>(o + at + 18, x); G::mx4<T>(o + at + 24, x); G::mx5<T>(o + at + 30, x);
    for (int i = 0; i < SPATIAL_ROW; i++) d_out[(size_t)k * SPATIAL_ROW + i] = o[i];
}

struct Ctx {
    hipStream_t *streams; G::robotModel<T> *d_robotModel; G::gridData<T> *hd; int K;
};

static int failed(const char *what) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, hipGetErrorString(grid_first_error()), grid_first_error_where(), grid_first_error_line());
    g_err = buf; grid_first_error() = hipSuccess; (void)hipGetLastError();
    return 1;
}
#define CHK(what) do { if (grid_first_error() != hipSuccess) return failed(what); } while (0)

extern "C" {

const char *as_last_error() { return g_err.c_str(); }
int as_num_joints() { return N; }
int as_spatial_row() { return SPATIAL_ROW; }

// The spatial-algebra device library on the GPU: xy [K][12] = (x | y) pairs, out [K][as_spatial_row()].  Returns 0 or 1 (HIP error).
int as_spatial(const T *xy, T alpha, int K, T *out) {
    T *d_xy = nullptr, *d_out = nullptr;
    gpuErrchk(hipMalloc((void **)&d_xy, sizeof(T) * 12 * (size_t)K));
    gpuErrchk(hipMalloc((void **)&d_out, sizeof(T) * SPATIAL_ROW * (size_t)K));
    CHK("as_spatial: alloc");
    gpuErrchk(hipMemcpy(d_xy, xy, sizeof(T) * 12 * (size_t)K, hipMemcpyHostToDevice));
    spatial_kernel<<<dim3((K + 63) / 64, 1, 1), dim3(64, 1, 1)>>>(d_out, d_xy, alpha, K);
    gpuErrchk(hipGetLastError());
    gpuErrchk(hipDeviceSynchronize());
    gpuErrchk(hipMemcpy(out, d_out, sizeof(T) * SPATIAL_ROW * (size_t)K, hipMemcpyDeviceToHost));
    (void)hipFree(d_xy); (void)hipFree(d_out);
    CHK("as_spatial");
    return 0;
}

// Run ONE named piece of the surface on K configurations.  q_qd_u [K][3n]; qdd [K][n] and Minv [K][n*n] (upper triangle)
// where the variant consumes them; `out` receives the variant's result rows (sizes in tests/test_api_surface.py).
// Returns 0, 1 (HIP error: as_last_error) or 2 (unknown name).
int as_run(const char *name_c, const T *q_qd_u, const T *qdd, const T *Minv, int K, T gravity, T *out) {
    const std::string name(name_c);
    Ctx c;
    c.K = K;
    c.streams = G::init_grid<T>();
    c.d_robotModel = G::init_robotModel<T>();
    c.hd = G::init_gridData<T>(K);
    CHK("init");
    G::gridData<T> *hd = c.hd;
    // fill every host input view the wrappers may read: q_qd_u, the compressed q_qd and q, qdd, Minv
    memcpy(hd->h_q_qd_u, q_qd_u, sizeof(T) * 3 * N * (size_t)K);
    for (int k = 0; k < K; k++) {
        memcpy(hd->h_q_qd + (size_t)k * 2 * N, q_qd_u + (size_t)k * 3 * N, sizeof(T) * 2 * N);
        memcpy(hd->h_q + (size_t)k * N, q_qd_u + (size_t)k * 3 * N, sizeof(T) * N);
    }
    if (qdd) memcpy(hd->h_qdd, qdd, sizeof(T) * N * (size_t)K);
    if (Minv) memcpy(hd->h_Minv, Minv, sizeof(T) * N * N * (size_t)K);
    const dim3 z(0, 0, 0);                       // "use the suggested launch shape"
    const dim3 blocks((K + 127) / 128, 1, 1), threads(128, 1, 1);      // and one explicit shape (two waves per block)
    const size_t nK = (size_t)K;
    int rc = 0;
    // device-resident inputs for the _compute_only / _launch variants (mode 0 does its own copies)
    const bool co = name.size() > 3 && name.compare(name.size() - 3, 3, "_co") == 0;
    const bool la = name.size() > 7 && name.compare(name.size() - 7, 7, "_launch") == 0;
    const std::string base = co ? name.substr(0, name.size() - 3) : (la ? name.substr(0, name.size() - 7) : name);
    if (co || la || base == "vaf_device" || base == "vaf_device_qdd" || base == "device_tier" || base == "inner_chain") {
        gpuErrchk(hipMemcpy(hd->d_q_qd_u, hd->h_q_qd_u, sizeof(T) * 3 * N * nK, hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(hd->d_q_qd, hd->h_q_qd, sizeof(T) * 2 * N * nK, hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(hd->d_q, hd->h_q, sizeof(T) * N * nK, hipMemcpyHostToDevice));
        if (qdd) gpuErrchk(hipMemcpy(hd->d_qdd, hd->h_qdd, sizeof(T) * N * nK, hipMemcpyHostToDevice));
        if (Minv) gpuErrchk(hipMemcpy(hd->d_Minv, hd->h_Minv, sizeof(T) * N * N * nK, hipMemcpyHostToDevice));
        CHK("upload");
    }
    hipStream_t s = c.streams[1];
    const T *d_result = nullptr; const T *h_result = nullptr; size_t count = 0;
#define HOST0(call, hptr, cnt) do { call; h_result = hd->hptr; count = (cnt); } while (0)
#define DEV(call, dptr, cnt) do { call; d_result = hd->dptr; count = (cnt); } while (0)
    if      (base == "id_cmem" && !co && !la)         HOST0((G::inverse_dynamics<T, false, true>(hd, c.d_robotModel, gravity, K, z, z, c.streams)), h_c, N * nK);
    else if (base == "id_qdd_cmem" && !co && !la)     HOST0((G::inverse_dynamics<T, true, true>(hd, c.d_robotModel, gravity, K, blocks, threads, c.streams)), h_c, N * nK);
    else if (base == "minv_cmem" && !co && !la)       HOST0((G::direct_minv<T, true>(hd, c.d_robotModel, K, z, z, c.streams)), h_Minv, N * N * nK);
    else if (base == "idgrad_cmem" && !co && !la)     HOST0((G::inverse_dynamics_gradient<T, false, true>(hd, c.d_robotModel, gravity, K, z, z, c.streams)), h_dc_du, 2 * N * N * nK);
    else if (base == "idgrad_qdd_cmem" && !co && !la) HOST0((G::inverse_dynamics_gradient<T, true, true>(hd, c.d_robotModel, gravity, K, blocks, threads, c.streams)), h_dc_du, 2 * N * N * nK);
    else if (base == "fdgrad_qddminv" && !co && !la)  HOST0((G::forward_dynamics_gradient<T, true>(hd, c.d_robotModel, gravity, K, z, z, c.streams)), h_df_du, 2 * N * N * nK);
    // ---- _compute_only (reference mode 2)
    else if (co && base == "id")             DEV((G::inverse_dynamics_compute_only<T, false, false>(hd, c.d_robotModel, gravity, K, z, z)), d_c, N * nK);
    else if (co && base == "id_qdd")         DEV((G::inverse_dynamics_compute_only<T, true, false>(hd, c.d_robotModel, gravity, K, blocks, threads)), d_c, N * nK);
    else if (co && base == "id_cmem")        DEV((G::inverse_dynamics_compute_only<T, false, true>(hd, c.d_robotModel, gravity, K, z, z)), d_c, N * nK);
    else if (co && base == "id_qdd_cmem")    DEV((G::inverse_dynamics_compute_only<T, true, true>(hd, c.d_robotModel, gravity, K, z, z)), d_c, N * nK);
    else if (co && base == "minv")           DEV((G::direct_minv_compute_only<T, false>(hd, c.d_robotModel, K, z, z)), d_Minv, N * N * nK);
    else if (co && base == "minv_cmem")      DEV((G::direct_minv_compute_only<T, true>(hd, c.d_robotModel, K, blocks, threads)), d_Minv, N * N * nK);
    else if (co && base == "fd")             DEV((G::forward_dynamics_compute_only<T>(hd, c.d_robotModel, gravity, K, z, z)), d_qdd, N * nK);
    else if (co && base == "idgrad")         DEV((G::inverse_dynamics_gradient_compute_only<T, false, false>(hd, c.d_robotModel, gravity, K, z, z)), d_dc_du, 2 * N * N * nK);
    else if (co && base == "idgrad_qdd")     DEV((G::inverse_dynamics_gradient_compute_only<T, true, false>(hd, c.d_robotModel, gravity, K, z, z)), d_dc_du, 2 * N * N * nK);
    else if (co && base == "idgrad_cmem")    DEV((G::inverse_dynamics_gradient_compute_only<T, false, true>(hd, c.d_robotModel, gravity, K, blocks, threads)), d_dc_du, 2 * N * N * nK);
    else if (co && base == "idgrad_qdd_cmem") DEV((G::inverse_dynamics_gradient_compute_only<T, true, true>(hd, c.d_robotModel, gravity, K, z, z)), d_dc_du, 2 * N * N * nK);
    else if (co && base == "fdgrad")         DEV((G::forward_dynamics_gradient_compute_only<T, false>(hd, c.d_robotModel, gravity, K, z, z)), d_df_du, 2 * N * N * nK);
    else if (co && base == "fdgrad_qddminv") DEV((G::forward_dynamics_gradient_compute_only<T, true>(hd, c.d_robotModel, gravity, K, blocks, threads)), d_df_du, 2 * N * N * nK);
    // ---- _launch (asynchronous on a caller stream; synchronised below)
    else if (la && base == "id")             DEV((G::inverse_dynamics_launch<T, false, false>(hd, c.d_robotModel, gravity, K, z, z, s)), d_c, N * nK);
    else if (la && base == "id_qdd_cmem")    DEV((G::inverse_dynamics_launch<T, true, true>(hd, c.d_robotModel, gravity, K, z, z, s)), d_c, N * nK);
    else if (la && base == "minv")           DEV((G::direct_minv_launch<T, false>(hd, c.d_robotModel, K, z, z, s)), d_Minv, N * N * nK);
    else if (la && base == "minv_cmem")      DEV((G::direct_minv_launch<T, true>(hd, c.d_robotModel, K, z, z, s)), d_Minv, N * N * nK);
    else if (la && base == "fd")             DEV((G::forward_dynamics_launch<T>(hd, c.d_robotModel, gravity, K, blocks, threads, s)), d_qdd, N * nK);
    else if (la && base == "idgrad")         DEV((G::inverse_dynamics_gradient_launch<T, false, false>(hd, c.d_robotModel, gravity, K, z, z, s)), d_dc_du, 2 * N * N * nK);
    else if (la && base == "idgrad_qdd_cmem") DEV((G::inverse_dynamics_gradient_launch<T, true, true>(hd, c.d_robotModel, gravity, K, z, z, s)), d_dc_du, 2 * N * N * nK);
    else if (la && base == "fdgrad")         DEV((G::forward_dynamics_gradient_launch<T, false>(hd, c.d_robotModel, gravity, K, z, z, s)), d_df_du, 2 * N * N * nK);
    else if (la && base == "fdgrad_qddminv") DEV((G::forward_dynamics_gradient_launch<T, true>(hd, c.d_robotModel, gravity, K, z, z, s)), d_df_du, 2 * N * N * nK);
    // ---- user kernels over the _device / _inner tiers
    else if (base == "vaf_device" || base == "vaf_device_qdd" || base == "device_tier" || base == "inner_chain") {
        const size_t row = (base == "device_tier") ? (size_t)(2 * N + 3 * N * N)
                         : (base == "inner_chain") ? (size_t)(N + N * N + 18 * N + 2 * N * N) : (size_t)(18 * N);
        T *d_tmp = nullptr;
        gpuErrchk(hipMalloc((void **)&d_tmp, sizeof(T) * row * nK));
        const dim3 b((K + 63) / 64), t(64);
        if (base == "device_tier")      device_tier_kernel<<<b, t, 0, s>>>(d_tmp, hd->d_q_qd_u, c.d_robotModel, gravity, K);
        else if (base == "inner_chain") inner_chain_kernel<<<b, t, 0, s>>>(d_tmp, hd->d_q_qd_u, c.d_robotModel, gravity, K);
        else vaf_device_kernel<<<b, t, 0, s>>>(d_tmp, hd->d_q_qd_u, base == "vaf_device_qdd" ? hd->d_qdd : nullptr, c.d_robotModel, gravity, K);
        gpuErrchk(hipGetLastError());
        gpuErrchk(hipStreamSynchronize(s));
        gpuErrchk(hipMemcpy(out, d_tmp, sizeof(T) * row * nK, hipMemcpyDeviceToHost));
        gpuErrchk(hipFree(d_tmp));
    }
    else rc = 2;
    if (rc == 0 && grid_first_error() == hipSuccess) {
        if (la) gpuErrchk(hipStreamSynchronize(s));
        if (d_result) gpuErrchk(hipMemcpy(out, d_result, sizeof(T) * count, hipMemcpyDeviceToHost));
        if (h_result) memcpy(out, h_result, sizeof(T) * count);
    }
    const bool bad = (grid_first_error() != hipSuccess);
    if (bad) failed(name_c);
    G::close_grid<T>(c.streams, c.d_robotModel, c.hd);
    grid_first_error() = hipSuccess;
    if (rc == 2) g_err = "unknown variant " + name;
    return bad ? 1 : rc;
}

}  // extern "C"
