// C-ABI shim over the generated header (one shared object per robot).  Declarations + the reference
// interface each entry point replaces: include/grid_capi.h.  Compile with
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DGRID_ERRORS_RETURN -DGRID_HEADER='"grid.hip.h"' -DGRID_NS=grid ...
#define GRID_ERRORS_RETURN 1
#include GRID_HEADER
#include <string.h>
#include <string>
#include "grid_capi.h"

#ifndef GRID_NS
#define GRID_NS grid
#endif
#ifndef GRID_ROBOT_NAME
#define GRID_ROBOT_NAME "robot"
#endif
namespace G = GRID_NS;
typedef float T;

#ifdef GRID_EXTERN_KERNELS
// kernels are instantiated one per translation unit (grid_kernel_inst.hip) and linked in
GRID_FOR_EACH_KERNEL_INST(extern template)
#endif

struct grid_handle {
    int device;
    G::robotModel<T> *d_robotModel;
    hipStream_t *streams;
    G::gridData<T> *hd_data;
    int max_timesteps;
    int split[5];   // per algorithm: 0 = auto, 1 = never split, S = force the S-way column-split kernel
    int pipeline[5];   // per algorithm: 0 = auto (single kernel), 1 = single kernel, 2 = two-pass (workspace) variant
    int coop[5];       // per algorithm: 0 = auto, 1 = never, 2 = always the tile-cooperative kernel (where generated), 3 = always its
                       // register-lean 8-wave variant (where generated)
    int wave[5];       // per algorithm: 0 = auto, 1 = never, 2 = always the wave-per-configuration kernel (where generated)
    int unsplit_regs[5];   // registers of the unsplit kernel (hipFuncGetAttributes at init): <= 256 means two waves share a SIMD
    T *d_workspace; size_t workspace_bytes; hipStream_t workspace_stream; bool workspace_busy;
};

static thread_local std::string g_last_error;

static int grid_fail(const char *what) {
    hipError_t e = grid_first_error();
    char buf[512];
    if (e != hipSuccess) {
        snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, hipGetErrorString(e), grid_first_error_where(), grid_first_error_line());
        grid_first_error() = hipSuccess;
        (void)hipGetLastError();
        g_last_error = buf;
        return (int)e;
    }
    g_last_error = what;
    return -1;
}
static inline int grid_check(const char *what) { return (grid_first_error() == hipSuccess) ? 0 : grid_fail(what); }
#define GRID_TRY(expr, what) do { hipError_t _e = (expr); if (_e != hipSuccess) { gpuAssert(_e, __FILE__, __LINE__); return grid_fail(what); } } while (0)

static void launch_shape(int K, int blocks, int threads, dim3 *b, dim3 *t) {
    dim3 ub(blocks > 0 ? blocks : 0, 1, 1), ut(threads > 0 ? threads : 0, 1, 1);
    if (blocks <= 0 && threads > 0) { ub = dim3((K + threads - 1) / threads > 0 ? (K + threads - 1) / threads : 1, 1, 1); }
    G::grid_launch_dims(ub, ut, K, b, t);
}

extern "C" {

const char *grid_robot_name(void) { return GRID_ROBOT_NAME; }
int grid_num_joints(void) { return G::NUM_JOINTS; }
int grid_topology_helpers_count(void) { return G::TOPOLOGY_HELPERS_COUNT; }
#define GRID_DTYPE_CAT2(ns) GRID_DTYPE_##ns
#define GRID_DTYPE_CAT(ns) GRID_DTYPE_CAT2(ns)
const char *grid_compute_dtype(void) { return GRID_DTYPE_CAT(GRID_NS); }
const char *grid_last_error(void) { return g_last_error.c_str(); }

int grid_constants(int *out, int count) {
    const int vals[10] = {G::NUM_JOINTS, G::ID_DYNAMIC_SHARED_MEM_COUNT, G::MINV_DYNAMIC_SHARED_MEM_COUNT, G::FD_DYNAMIC_SHARED_MEM_COUNT,
                          G::ID_DU_DYNAMIC_SHARED_MEM_COUNT, G::FD_DU_DYNAMIC_SHARED_MEM_COUNT, G::ID_DU_MAX_SHARED_MEM_COUNT,
                          G::FD_DU_MAX_SHARED_MEM_COUNT, G::SUGGESTED_THREADS, G::SUGGESTED_MAX_BLOCKS};
    if (out == nullptr || count < 0) { g_last_error = "grid_constants: bad arguments"; return -1; }
    for (int i = 0; i < count && i < 10; i++) out[i] = vals[i];
    return 0;
}

int grid_init(int device, grid_handle **out) {
    if (out == nullptr) { g_last_error = "grid_init: out is NULL"; return -1; }
    *out = nullptr;
    int ndev = 0;
    GRID_TRY(hipGetDeviceCount(&ndev), "grid_init: hipGetDeviceCount");
    if (device < 0 || device >= ndev) { g_last_error = "grid_init: no such device"; return -1; }
    GRID_TRY(hipSetDevice(device), "grid_init: hipSetDevice");
    grid_handle *h = new grid_handle();
    h->device = device; h->hd_data = nullptr; h->max_timesteps = 0;
    for (int a = 0; a < 5; a++) { h->split[a] = 0; h->pipeline[a] = 0; h->coop[a] = 0; h->wave[a] = 0; }
    h->d_workspace = nullptr; h->workspace_bytes = 0; h->workspace_stream = nullptr; h->workspace_busy = false;
    for (int a = 0; a < 5; a++) { int attr[4] = {0, 0, 0, 0}; h->unsplit_regs[a] = (grid_kernel_attributes(a, 0, attr) == 0) ? attr[0] : 512; }
    h->d_robotModel = G::init_robotModel<T>();
    h->streams = G::init_grid<T>();
    if (int rc = grid_check("grid_init")) { delete h; return rc; }
    *out = h;
    return 0;
}

int grid_alloc(grid_handle *h, int max_timesteps) {
    if (h == nullptr || max_timesteps <= 0) { g_last_error = "grid_alloc: bad arguments"; return -1; }
    GRID_TRY(hipSetDevice(h->device), "grid_alloc: hipSetDevice");
    if (h->hd_data != nullptr) {
        G::close_grid<T>(nullptr, nullptr, h->hd_data);
        h->hd_data = nullptr; h->max_timesteps = 0;
    }
    h->hd_data = G::init_gridData<T>(max_timesteps);
    if (int rc = grid_check("grid_alloc")) return rc;
    h->max_timesteps = max_timesteps;
    return 0;
}

int grid_close(grid_handle *h) {
    if (h == nullptr) return 0;
    (void)hipSetDevice(h->device);
    G::close_grid<T>(h->streams, h->d_robotModel, h->hd_data);
    if (h->d_workspace != nullptr) (void)hipFree(h->d_workspace);
    int rc = grid_check("grid_close");
    delete h;
    return rc;
}

int grid_read_model(grid_handle *h, float *h_XImats, int *h_topology) {
    if (h == nullptr) { g_last_error = "grid_read_model: NULL handle"; return -1; }
    GRID_TRY(hipSetDevice(h->device), "grid_read_model: hipSetDevice");
    G::robotModel<T> hm;
    GRID_TRY(hipMemcpy(&hm, h->d_robotModel, sizeof(hm), hipMemcpyDeviceToHost), "grid_read_model: struct");
    if (h_XImats) GRID_TRY(hipMemcpy(h_XImats, hm.d_XImats, G::XIMATS_MODEL_COUNT * sizeof(T), hipMemcpyDeviceToHost), "grid_read_model: XImats");
    if (h_topology && G::TOPOLOGY_HELPERS_COUNT > 0)
        GRID_TRY(hipMemcpy(h_topology, hm.d_topology_helpers, G::TOPOLOGY_HELPERS_COUNT * sizeof(int), hipMemcpyDeviceToHost), "grid_read_model: topology");
    return 0;
}

static int host_prep(grid_handle *h, const float *h_q_qd_u, int K, const char *who) {
    if (h == nullptr || h_q_qd_u == nullptr || K <= 0) { g_last_error = std::string(who) + ": bad arguments"; return -1; }
    if (h->hd_data == nullptr || K > h->max_timesteps) { g_last_error = std::string(who) + ": call grid_alloc(h, >= num_timesteps) first"; return -1; }
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) { gpuAssert(e, __FILE__, __LINE__); return grid_fail(who); }
    memcpy(h->hd_data->h_q_qd_u, h_q_qd_u, sizeof(T) * 3 * G::NUM_JOINTS * (size_t)K);
    return 0;
}

int grid_inverse_dynamics(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, float *h_c, int K, float gravity) {
    if (int rc = host_prep(h, h_q_qd_u, K, "grid_inverse_dynamics")) return rc;
    if (h_c == nullptr) { g_last_error = "grid_inverse_dynamics: h_c is NULL"; return -1; }
    const dim3 z(0, 0, 0);
    if (h_qdd) { memcpy(h->hd_data->h_qdd, h_qdd, sizeof(T) * G::NUM_JOINTS * (size_t)K);
                 G::inverse_dynamics<T, true, false>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams); }
    else       { G::inverse_dynamics<T, false, false>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams); }
    if (int rc = grid_check("grid_inverse_dynamics")) return rc;
    memcpy(h_c, h->hd_data->h_c, sizeof(T) * G::NUM_JOINTS * (size_t)K);
    return 0;
}

int grid_direct_minv(grid_handle *h, const float *h_q_qd_u, float *h_Minv, int K) {
    if (int rc = host_prep(h, h_q_qd_u, K, "grid_direct_minv")) return rc;
    if (h_Minv == nullptr) { g_last_error = "grid_direct_minv: h_Minv is NULL"; return -1; }
    const dim3 z(0, 0, 0);
    G::direct_minv<T, false>(h->hd_data, h->d_robotModel, K, z, z, h->streams);
    if (int rc = grid_check("grid_direct_minv")) return rc;
    memcpy(h_Minv, h->hd_data->h_Minv, sizeof(T) * G::NUM_JOINTS * G::NUM_JOINTS * (size_t)K);
    return 0;
}

int grid_forward_dynamics(grid_handle *h, const float *h_q_qd_u, float *h_qdd, int K, float gravity) {
    if (int rc = host_prep(h, h_q_qd_u, K, "grid_forward_dynamics")) return rc;
    if (h_qdd == nullptr) { g_last_error = "grid_forward_dynamics: h_qdd is NULL"; return -1; }
    const dim3 z(0, 0, 0);
    G::forward_dynamics<T>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams);
    if (int rc = grid_check("grid_forward_dynamics")) return rc;
    memcpy(h_qdd, h->hd_data->h_qdd, sizeof(T) * G::NUM_JOINTS * (size_t)K);
    return 0;
}

int grid_inverse_dynamics_gradient(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, float *h_dc_du, int K, float gravity) {
    if (int rc = host_prep(h, h_q_qd_u, K, "grid_inverse_dynamics_gradient")) return rc;
    if (h_dc_du == nullptr) { g_last_error = "grid_inverse_dynamics_gradient: h_dc_du is NULL"; return -1; }
    const dim3 z(0, 0, 0);
    if (h_qdd) { memcpy(h->hd_data->h_qdd, h_qdd, sizeof(T) * G::NUM_JOINTS * (size_t)K);
                 G::inverse_dynamics_gradient<T, true, false>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams); }
    else       { G::inverse_dynamics_gradient<T, false, false>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams); }
    if (int rc = grid_check("grid_inverse_dynamics_gradient")) return rc;
    memcpy(h_dc_du, h->hd_data->h_dc_du, sizeof(T) * 2 * G::NUM_JOINTS * G::NUM_JOINTS * (size_t)K);
    return 0;
}

int grid_forward_dynamics_gradient(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, const float *h_Minv, float *h_df_du,
                                   int K, float gravity) {
    if (int rc = host_prep(h, h_q_qd_u, K, "grid_forward_dynamics_gradient")) return rc;
    if (h_df_du == nullptr) { g_last_error = "grid_forward_dynamics_gradient: h_df_du is NULL"; return -1; }
    if ((h_qdd == nullptr) != (h_Minv == nullptr)) { g_last_error = "grid_forward_dynamics_gradient: pass both h_qdd and h_Minv or neither"; return -1; }
    const dim3 z(0, 0, 0);
    if (h_qdd) {
        memcpy(h->hd_data->h_qdd, h_qdd, sizeof(T) * G::NUM_JOINTS * (size_t)K);
        memcpy(h->hd_data->h_Minv, h_Minv, sizeof(T) * G::NUM_JOINTS * G::NUM_JOINTS * (size_t)K);
        G::forward_dynamics_gradient<T, true>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams);
    } else {
        G::forward_dynamics_gradient<T, false>(h->hd_data, h->d_robotModel, gravity, K, z, z, h->streams);
    }
    if (int rc = grid_check("grid_forward_dynamics_gradient")) return rc;
    memcpy(h_df_du, h->hd_data->h_df_du, sizeof(T) * 2 * G::NUM_JOINTS * G::NUM_JOINTS * (size_t)K);
    return 0;
}

// ---- device-pointer launches ------------------------------------------------------------------
// Column-split choice (gradient kernels, default input variants only).  Splitting trades repeated prefix work (X(q), Minv,
// RNEA) for more wavefronts.  Measured on MI355X (iiwa-7 FD gradient, tools/exp_splits.py): it pays while the batch leaves
// SIMDs idle, so pick the largest S with tiles*S <= 4 x 256 (one wave on every SIMD; the finer splits are compiled without
// a register cap and a second wave on a SIMD would serialise behind the first).  K=16384: 17.7 us (S=1), 13.6 (S=2),
// 12.4 (S=3), 11.5 (S=4), 18.3 (S=5).  For batches that already fill the chip what counts is two waves per SIMD: the unsplit
// kernel when it fits 256 registers (iiwa-7 dFD since the kernels are branch-free: K=262144 53 us against 77 us for the 2-way
// split, 1M: 202 against 321 us), otherwise the register-capped 2-way split (round 1: unsplit needed 280 registers, 319 -> 285 us).
static const int GRID_CUS = 256;   // MI355X
static int available_splits(int alg, const int **list) {
    if (alg == GRID_ALG_FD_DU) { *list = G::FD_DU_SPLITS; return G::FD_DU_NUM_SPLITS; }
    if (alg == GRID_ALG_ID_DU) { *list = G::ID_DU_SPLITS; return G::ID_DU_NUM_SPLITS; }
    *list = nullptr; return 0;
}
static int effective_split(const grid_handle *h, int alg, int K) {
    const int *list; const int n = available_splits(alg, &list);
    if (n == 0 || alg < 0 || alg > 4) return 1;
    const int want = h->split[alg];
    if (want == 1) return 1;
    if (want > 1) { for (int i = 0; i < n; i++) if (list[i] == want) return want; return 1; }
    const int tiles = (K + G::GRID_WAVE_SIZE - 1) / G::GRID_WAVE_SIZE;
    int best = 1;
    // Large robots (column groups of the recomputing schedule, not register-capped, no repeated prefix for dID): measured on
    // Atlas-30 (tools/split_sweep.py, profiles/r01/sweep_atlas30_split_recompute.txt) the finest split wins for dID at EVERY
    // batch (K=16384: 160 -> 76 us, 65536: 392 -> 325, 262144: 1367 -> 1252: shorter waves overlap their output stores with
    // other waves' arithmetic), and for dFD (every group repeats Minv) up to 768 tiles (K=16384: 261 -> 125 us,
    // 32768: 373 -> 253, 49152: 476 -> 421; 65536: 528 -> 543, so not beyond).
    if (G::NUM_JOINTS > 12) {
        if (alg == GRID_ALG_FD_DU && tiles > 3 * GRID_CUS) return 1;
        for (int i = 0; i < n; i++) if (list[i] > best) best = list[i];
        return best;
    }
    for (int i = 0; i < n; i++) if ((long long)tiles * list[i] <= 4LL * GRID_CUS && list[i] > best) best = list[i];
    if (best == 1 && G::NUM_JOINTS <= 12 && h->unsplit_regs[alg] > 256) { for (int i = 0; i < n; i++) if (list[i] == 2) best = 2; }
    return best;
}

// Tile-cooperative kernel (forward-dynamics gradient): 4 waves share a tile, the prefix (Minv | RNEA) is computed once per tile and
// Minv is read from LDS instead of being held in registers.  The automatic choice is a property of the generated kernels and comes
// from the header (FD_DU_COOP_AUTO_MIN_TILES, with the measurements behind it): every batch size for large robots in fp32, full
// chips only in the mixed arithmetic, small robots only on request.
static bool coop_available(int alg) { return alg == GRID_ALG_FD_DU && G::FD_DU_COOP_WAVES > 0; }
static bool lean_available(int alg) {
    return (alg == GRID_ALG_FD_DU && G::FD_DU_LEAN_WAVES > 0) || (alg == GRID_ALG_ID_DU && G::ID_DU_LEAN_WAVES > 0) || (alg == GRID_ALG_FD && G::FD_LEAN_WAVES > 0)
        || (alg == GRID_ALG_MINV && G::MINV_LEAN_WAVES > 0);
}
static int lean_auto_min_tiles(int alg) {
    return alg == GRID_ALG_FD_DU ? G::FD_DU_LEAN_AUTO_MIN_TILES : (alg == GRID_ALG_ID_DU ? G::ID_DU_LEAN_AUTO_MIN_TILES : (alg == GRID_ALG_FD ? G::FD_LEAN_AUTO_MIN_TILES
         : (alg == GRID_ALG_MINV ? G::MINV_LEAN_AUTO_MIN_TILES : 0)));
}
// 0: neither, 1: the 4-wave tile-cooperative kernel, 2: its register-lean 8-wave variant (two waves per SIMD, <= 256 registers each)
static int coop_variant(const grid_handle *h, int alg, int K, const float *d_qdd, const float *d_Minv) {
    if (alg < 0 || alg > 4 || d_qdd != nullptr || d_Minv != nullptr || h->coop[alg] == 1) return 0;
    if (h->coop[alg] == 3) return lean_available(alg) ? 2 : 0;
    if (h->coop[alg] == 2) return coop_available(alg) ? 1 : 0;
    if (h->split[alg] != 0 || h->pipeline[alg] == 2 || h->wave[alg] == 2) return 0;              // an explicit choice of another variant wins
    const int tiles = (K + G::GRID_WAVE_SIZE - 1) / G::GRID_WAVE_SIZE;
    const int lean_max = (alg == GRID_ALG_FD) ? G::FD_LEAN_AUTO_MAX_TILES : (alg == GRID_ALG_MINV) ? G::MINV_LEAN_AUTO_MAX_TILES : 0;       // (forward dynamics: beyond two tiles per CU the lane kernel wins)
    if (lean_available(alg) && lean_auto_min_tiles(alg) > 0 && tiles >= lean_auto_min_tiles(alg) && (lean_max == 0 || tiles <= lean_max)) return 2;
    if (coop_available(alg) && G::FD_DU_COOP_AUTO_MIN_TILES > 0 && tiles >= G::FD_DU_COOP_AUTO_MIN_TILES) return 1;      // (the generated header knows: see its comment)
    return 0;
}
static bool use_coop(const grid_handle *h, int alg, int K, const float *d_qdd, const float *d_Minv) { return coop_variant(h, alg, K, d_qdd, d_Minv) != 0; }

// Wave-per-configuration kernels (all five algorithms): one block per configuration, the lanes of a wavefront are the gradient
// columns / Minv columns / joints of a group of base-rooted trees.  A configuration then takes as long as its longest group's chain
// ON 64 LANES, not as long as the whole chain on one lane: the small-batch path.  Automatic up to <ALG>_WAVE_AUTO_MAX_K configurations
// (generated header: large robots; the measurements are quoted there); an explicit choice of another variant wins.
static int wave_auto_max_k_header(int alg) {
    switch (alg) {
    case GRID_ALG_ID: return G::ID_WAVE_AUTO_MAX_K;
    case GRID_ALG_MINV: return G::MINV_WAVE_AUTO_MAX_K;
    case GRID_ALG_FD: return G::FD_WAVE_AUTO_MAX_K;
    case GRID_ALG_ID_DU: return G::ID_DU_WAVE_AUTO_MAX_K;
    case GRID_ALG_FD_DU: return G::FD_DU_WAVE_AUTO_MAX_K;
    default: return 0;
    }
}
// ... capped where the library has a register-lean tile-cooperative kernel that is faster from a smaller batch on (<ALG>_LEAN_WAVE_MAX_K)
static int wave_auto_max_k(int alg) {
    const int k = wave_auto_max_k_header(alg);
    const int cap = (alg == GRID_ALG_FD_DU && G::FD_DU_LEAN_WAVES > 0) ? G::FD_DU_LEAN_WAVE_MAX_K
                  : (alg == GRID_ALG_ID_DU && G::ID_DU_LEAN_WAVES > 0) ? G::ID_DU_LEAN_WAVE_MAX_K
                  : (alg == GRID_ALG_FD && G::FD_LEAN_WAVES > 0) ? G::FD_LEAN_WAVE_MAX_K
                  : (alg == GRID_ALG_MINV && G::MINV_LEAN_WAVES > 0) ? G::MINV_LEAN_WAVE_MAX_K : 0;
    return (cap > 0 && cap < k) ? cap : k;
}
static bool wave_available(int alg) { return alg >= 0 && alg <= 4 && G::FD_DU_WAVE_WAVES > 0; }
static bool use_wave(const grid_handle *h, int alg, int K, const float *d_qdd, const float *d_Minv, int blocks = 0, int threads = 0) {
    if (!wave_available(alg) || d_Minv != nullptr || h->wave[alg] == 1) return false;
    if (alg == GRID_ALG_FD_DU && d_qdd != nullptr) return false;       // (precomputed qdd/Minv: the lane-per-configuration kernel)
    if (h->wave[alg] == 2) return true;
    if (h->split[alg] != 0 || h->pipeline[alg] == 2 || h->coop[alg] >= 2) return false;
    // A caller that passes a launch shape means blocks of `threads` CONFIGURATIONS (the reference's <<<block_dimms, thread_dimms>>>,
    // lane-per-configuration here); in the wave-per-configuration kernels a block IS one configuration, so the same numbers would mean
    // blocks*1 configurations in flight, each block walking K/blocks of them serially.  The automatic choice therefore keeps to the
    // launch shape's meaning: an explicit shape gets the lane-per-configuration kernels (grid_set_wave(..., 2) takes `blocks` as the
    // number of configurations in flight on purpose).
    if (blocks > 0 || threads > 0) return false;
    return wave_auto_max_k(alg) > 0 && K <= wave_auto_max_k(alg);
}

// Two-pass (workspace) variants: generated for robots whose gradient working set exceeds the register file.
static int workspace_count(int alg) {
    return alg == GRID_ALG_FD_DU ? G::FD_DU_WORKSPACE_COUNT : (alg == GRID_ALG_ID_DU ? G::ID_DU_WORKSPACE_COUNT : 0);
}
static bool use_pipeline(const grid_handle *h, int alg, const float *d_qdd, const float *d_Minv) {
    if (alg < 0 || alg > 4 || workspace_count(alg) == 0 || d_Minv != nullptr) return false;
    if (alg == GRID_ALG_FD_DU && d_qdd != nullptr) return false;
    return h->pipeline[alg] == 2;      // auto = single kernel: since the recomputing column schedule it is the faster one
                                       // (Atlas-30 K=65536: 372 vs 531 us dID, 561 vs 799 us dFD); the two-pass variant is opt-in
}
static int ensure_workspace(grid_handle *h, int alg, int K, hipStream_t s) {
    // one workspace per handle: work queued on another stream may still be reading it
    if (h->workspace_busy && h->workspace_stream != s) GRID_TRY(hipStreamSynchronize(h->workspace_stream), "workspace: stream hand-over");
    h->workspace_stream = s; h->workspace_busy = true;
    const size_t need = (size_t)workspace_count(alg) * (size_t)((K + G::GRID_WAVE_SIZE - 1) / G::GRID_WAVE_SIZE) * G::GRID_WAVE_SIZE * sizeof(T);
    if (need <= h->workspace_bytes) return 0;
    if (h->d_workspace != nullptr) { GRID_TRY(hipDeviceSynchronize(), "workspace: sync"); GRID_TRY(hipFree(h->d_workspace), "workspace: free"); h->d_workspace = nullptr; h->workspace_bytes = 0; }
    GRID_TRY(hipMalloc((void **)&h->d_workspace, need), "workspace: hipMalloc");
    h->workspace_bytes = need;
    return 0;
}

static int launch_alg(grid_handle *h, int alg, float *d_out, const float *d_in, int stride, const float *d_qdd, const float *d_Minv,
                      int K, float gravity, int blocks, int threads, hipStream_t s) {
    dim3 b, t;
    launch_shape(K, blocks, threads, &b, &t);
    if (use_wave(h, alg, K, d_qdd, d_Minv, blocks, threads)) {
        const int nb = blocks > 0 ? blocks : 0;       // (reached with an explicit shape only under grid_set_wave(..., 2): configurations in flight)
        switch (alg) {
        case GRID_ALG_ID:    G::inverse_dynamics_wave_launch<T>(d_out, d_in, stride, d_qdd, h->d_robotModel, gravity, K, nb, s); break;
        case GRID_ALG_MINV:  G::direct_minv_wave_launch<T>(d_out, d_in, stride, h->d_robotModel, K, nb, s); break;
        case GRID_ALG_FD:    G::forward_dynamics_wave_launch<T>(d_out, d_in, stride, h->d_robotModel, gravity, K, nb, s); break;
        case GRID_ALG_ID_DU: G::inverse_dynamics_gradient_wave_launch<T>(d_out, d_in, stride, d_qdd, h->d_robotModel, gravity, K, nb, s); break;
        default:             G::forward_dynamics_gradient_wave_launch<T>(d_out, d_in, stride, h->d_robotModel, gravity, K, nb, s); break;
        }
        return grid_check("kernel launch (wave-per-configuration)");
    }
    if (use_coop(h, alg, K, d_qdd, d_Minv)) {
        // an explicit launch shape means blocks x threads configurations in flight: that many tiles of 64 (one block of the kernel each)
        int tile_blocks = 0;
        if (blocks > 0) { const long long cfgs = (long long)blocks * (threads > 0 ? threads : G::GRID_WAVE_SIZE);
                          tile_blocks = (int)((cfgs + G::GRID_WAVE_SIZE - 1) / G::GRID_WAVE_SIZE); }
        if (alg == GRID_ALG_ID_DU) G::inverse_dynamics_gradient_lean_launch<T>(d_out, d_in, stride, h->d_robotModel, gravity, K, tile_blocks, s);
        else if (alg == GRID_ALG_FD) G::forward_dynamics_lean_launch<T>(d_out, d_in, stride, h->d_robotModel, gravity, K, tile_blocks, s);
        else if (alg == GRID_ALG_MINV) G::direct_minv_lean_launch<T>(d_out, d_in, stride, h->d_robotModel, K, tile_blocks, s);
        else if (coop_variant(h, alg, K, d_qdd, d_Minv) == 2) G::forward_dynamics_gradient_lean_launch<T>(d_out, d_in, stride, h->d_robotModel, gravity, K, tile_blocks, s);
        else G::forward_dynamics_gradient_coop_launch<T>(d_out, d_in, stride, h->d_robotModel, gravity, K, tile_blocks, s);
        return grid_check("kernel launch (tile-cooperative)");
    }
    if (use_pipeline(h, alg, d_qdd, d_Minv)) {
        if (int rc = ensure_workspace(h, alg, K, s)) return rc;
        if (alg == GRID_ALG_FD_DU) G::forward_dynamics_gradient_pipeline_launch<T>(d_out, d_in, stride, nullptr, h->d_workspace, h->d_robotModel, gravity, K, b, t, s);
        else                       G::inverse_dynamics_gradient_pipeline_launch<T>(d_out, d_in, stride, d_qdd, h->d_workspace, h->d_robotModel, gravity, K, b, t, s);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { gpuAssert(e, __FILE__, __LINE__); return grid_fail("kernel launch (two-pass)"); }
        return 0;
    }
    if ((alg == GRID_ALG_FD_DU || alg == GRID_ALG_ID_DU) && d_qdd == nullptr && d_Minv == nullptr) {
        const int S = effective_split(h, alg, K);
        if (S > 1) {
            // tiles_in_flight x S wavefronts, packed into blocks of 64*S threads (capped at GRID_MAX_THREADS) unless the caller chose a
            // block size: the column groups of a tile are then the waves of one block -- one CU, one XCD's L2 for the pieces of
            // the tile's output rows (profiles/r03: WRITE_SIZE per launch)
            const int tiles = (K + G::GRID_WAVE_SIZE - 1) / G::GRID_WAVE_SIZE;
            int tiles_in_flight = (blocks > 0) ? (int)(b.x * b.y * b.z) : tiles;
            if (tiles_in_flight > tiles) tiles_in_flight = tiles;
            if (tiles_in_flight * S > G::SUGGESTED_MAX_BLOCKS * 4) tiles_in_flight = (G::SUGGESTED_MAX_BLOCKS * 4) / S;
            // ... when the launch has more waves than the chip has CUs and every wave is resident at once (one per SIMD).  Fewer waves:
            // single-wave blocks, one per CU -- a wave alone on a CU is 4-6 % faster (iiwa-7 dFD x4 at K = 4096: 8.7 vs 9.1 us); more:
            // single-wave blocks again, which the dispatcher back-fills SIMD by SIMD (a 4-wave block of a 512-register kernel needs a
            // whole free CU).  Measured, iiwa-7 K = 16384: dFD x4 9.9 vs 11.6 us, dID x4 7.1 vs 9.9 us (profiles/r03/exp_iiwa7_column_sets.txt)
            const long long nwaves = (long long)tiles_in_flight * S;
            const dim3 tb = (threads > 0) ? t : ((nwaves > GRID_CUS && nwaves <= 4LL * GRID_CUS) ? dim3(0, 1, 1) : dim3(G::GRID_WAVE_SIZE, 1, 1));
            bool ok = (alg == GRID_ALG_FD_DU)
                ? G::forward_dynamics_gradient_split_launch<T>(S, d_out, d_in, stride, h->d_robotModel, gravity, K, tiles_in_flight, tb, s)
                : G::inverse_dynamics_gradient_split_launch<T>(S, d_out, d_in, stride, h->d_robotModel, gravity, K, tiles_in_flight, tb, s);
            if (ok) {
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) { gpuAssert(e, __FILE__, __LINE__); return grid_fail("kernel launch (split)"); }
                return 0;
            }
        }
    }
    switch (alg) {
    case GRID_ALG_ID: {
        const size_t lds = G::grid_lds_bytes<T>(t, G::ID_DYNAMIC_SHARED_MEM_COUNT / (G::GRID_MAX_THREADS / G::GRID_WAVE_SIZE));
        if (d_qdd) G::inverse_dynamics_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, d_qdd, h->d_robotModel, gravity, K);
        else       G::inverse_dynamics_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, h->d_robotModel, gravity, K);
        break; }
    case GRID_ALG_MINV: {
        const size_t lds = G::grid_lds_bytes<T>(t, G::MINV_DYNAMIC_SHARED_MEM_COUNT / (G::GRID_MAX_THREADS / G::GRID_WAVE_SIZE));
        G::direct_minv_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, h->d_robotModel, K);
        break; }
    case GRID_ALG_FD: {
        const size_t lds = G::grid_lds_bytes<T>(t, G::FD_DYNAMIC_SHARED_MEM_COUNT / (G::GRID_MAX_THREADS / G::GRID_WAVE_SIZE));
        G::forward_dynamics_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, h->d_robotModel, gravity, K);
        break; }
    case GRID_ALG_ID_DU: {
        const size_t lds = G::grid_lds_bytes<T>(t, G::ID_DU_DYNAMIC_SHARED_MEM_COUNT / (G::GRID_MAX_THREADS / G::GRID_WAVE_SIZE));
        if (d_qdd) G::inverse_dynamics_gradient_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, d_qdd, h->d_robotModel, gravity, K);
        else       G::inverse_dynamics_gradient_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, h->d_robotModel, gravity, K);
        break; }
    case GRID_ALG_FD_DU: {
        const size_t lds = G::grid_lds_bytes<T>(t, G::FD_DU_DYNAMIC_SHARED_MEM_COUNT / (G::GRID_MAX_THREADS / G::GRID_WAVE_SIZE));
        if (d_qdd && d_Minv) G::forward_dynamics_gradient_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, d_qdd, d_Minv, h->d_robotModel, gravity, K);
        else                 G::forward_dynamics_gradient_kernel<T><<<b, t, lds, s>>>(d_out, d_in, stride, h->d_robotModel, gravity, K);
        break; }
    default:
        g_last_error = "unknown algorithm id"; return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpuAssert(e, __FILE__, __LINE__); return grid_fail("kernel launch"); }
    return 0;
}

static int dev_prep(grid_handle *h, const void *d_out, const void *d_in, int K, const char *who) {
    if (h == nullptr || d_out == nullptr || d_in == nullptr || K <= 0) { g_last_error = std::string(who) + ": bad arguments"; return -1; }
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) { gpuAssert(e, __FILE__, __LINE__); return grid_fail(who); }
    return 0;
}
// stream == NULL is HIP's default stream: ordered with everything the caller has queued on it (and on every blocking stream), so a
// caller that fills its buffers on the default stream and launches with NULL needs no synchronisation.  The handle's own
// non-blocking streams (init_grid: three, descending priority, NOT ordered with the default stream) serve the host-buffer wrappers
// and are handed out by grid_stream() to callers that want them.
static inline hipStream_t pick_stream(grid_handle *h, void *stream) { (void)h; return (hipStream_t)stream; }

int grid_inverse_dynamics_device(grid_handle *h, float *d_c, const float *d_q_qd, int stride_q_qd, const float *d_qdd,
                                 int K, float gravity, int blocks, int threads, void *stream) {
    if (int rc = dev_prep(h, d_c, d_q_qd, K, "grid_inverse_dynamics_device")) return rc;
    return launch_alg(h, GRID_ALG_ID, d_c, d_q_qd, stride_q_qd, d_qdd, nullptr, K, gravity, blocks, threads, pick_stream(h, stream));
}
int grid_direct_minv_device(grid_handle *h, float *d_Minv, const float *d_q, int stride_q, int K, int blocks, int threads, void *stream) {
    if (int rc = dev_prep(h, d_Minv, d_q, K, "grid_direct_minv_device")) return rc;
    return launch_alg(h, GRID_ALG_MINV, d_Minv, d_q, stride_q, nullptr, nullptr, K, 0.0f, blocks, threads, pick_stream(h, stream));
}
int grid_forward_dynamics_device(grid_handle *h, float *d_qdd, const float *d_q_qd_u, int stride_q_qd_u,
                                 int K, float gravity, int blocks, int threads, void *stream) {
    if (int rc = dev_prep(h, d_qdd, d_q_qd_u, K, "grid_forward_dynamics_device")) return rc;
    return launch_alg(h, GRID_ALG_FD, d_qdd, d_q_qd_u, stride_q_qd_u, nullptr, nullptr, K, gravity, blocks, threads, pick_stream(h, stream));
}
int grid_inverse_dynamics_gradient_device(grid_handle *h, float *d_dc_du, const float *d_q_qd, int stride_q_qd, const float *d_qdd,
                                          int K, float gravity, int blocks, int threads, void *stream) {
    if (int rc = dev_prep(h, d_dc_du, d_q_qd, K, "grid_inverse_dynamics_gradient_device")) return rc;
    return launch_alg(h, GRID_ALG_ID_DU, d_dc_du, d_q_qd, stride_q_qd, d_qdd, nullptr, K, gravity, blocks, threads, pick_stream(h, stream));
}
int grid_forward_dynamics_gradient_device(grid_handle *h, float *d_df_du, const float *d_q_qd_u, int stride_q_qd_u,
                                          const float *d_qdd, const float *d_Minv, int K, float gravity, int blocks, int threads, void *stream) {
    if (int rc = dev_prep(h, d_df_du, d_q_qd_u, K, "grid_forward_dynamics_gradient_device")) return rc;
    if ((d_qdd == nullptr) != (d_Minv == nullptr)) { g_last_error = "grid_forward_dynamics_gradient_device: pass both d_qdd and d_Minv or neither"; return -1; }
    return launch_alg(h, GRID_ALG_FD_DU, d_df_du, d_q_qd_u, stride_q_qd_u, d_qdd, d_Minv, K, gravity, blocks, threads, pick_stream(h, stream));
}

int grid_rollout_row_count(void) { return G::ROLLOUT_ROW_COUNT; }
int grid_forward_dynamics_gradient_rollout_device(grid_handle *h, float *d_traj, const float *d_x0, const float *d_u_traj,
                                                  int K, int num_steps, float dt, float gravity, int blocks, int threads, void *stream) {
    if (int rc = dev_prep(h, d_traj, d_x0, K, "grid_forward_dynamics_gradient_rollout_device")) return rc;
    if (d_u_traj == nullptr || num_steps <= 0) { g_last_error = "grid_forward_dynamics_gradient_rollout_device: bad arguments"; return -1; }
    dim3 b, t;
    launch_shape(K, blocks, threads, &b, &t);
    G::forward_dynamics_gradient_rollout_launch<T>(d_traj, d_x0, d_u_traj, dt, h->d_robotModel, gravity, K, num_steps, b, t, pick_stream(h, stream));
    return grid_check("grid_forward_dynamics_gradient_rollout_device");
}

int grid_splits(int alg, int *out, int count) {
    const int *list; const int n = available_splits(alg, &list);
    for (int i = 0; i < n && i < count && out != nullptr; i++) out[i] = list[i];
    return n;
}
int grid_set_split(grid_handle *h, int alg, int split) {
    if (h == nullptr || alg < 0 || alg > 4 || split < 0) { g_last_error = "grid_set_split: bad arguments"; return -1; }
    if (split > 1) {
        const int *list; const int n = available_splits(alg, &list);
        bool found = false;
        for (int i = 0; i < n; i++) found = found || (list[i] == split);
        if (!found) { g_last_error = "grid_set_split: this split was not generated for this robot/algorithm"; return -1; }
    }
    h->split[alg] = split;
    return 0;
}
int grid_get_split(grid_handle *h, int alg, int num_timesteps) {
    if (h == nullptr || alg < 0 || alg > 4) return -1;
    return effective_split(h, alg, num_timesteps);
}

int grid_coop_available(int alg) { return coop_available(alg) ? 1 : 0; }
int grid_lean_available(int alg) { return lean_available(alg) ? 1 : 0; }
int grid_set_coop(grid_handle *h, int alg, int mode) {
    if (h == nullptr || alg < 0 || alg > 4 || mode < 0 || mode > 3) { g_last_error = "grid_set_coop: bad arguments"; return -1; }
    if (mode == 2 && !coop_available(alg)) { g_last_error = "grid_set_coop: no tile-cooperative kernel was generated for this robot/algorithm"; return -1; }
    if (mode == 3 && !lean_available(alg)) { g_last_error = "grid_set_coop: no register-lean tile-cooperative kernel was generated for this robot/algorithm"; return -1; }
    h->coop[alg] = mode;
    return 0;
}
int grid_get_coop(grid_handle *h, int alg, int num_timesteps) {
    if (h == nullptr || alg < 0 || alg > 4) return -1;
    return coop_variant(h, alg, num_timesteps, nullptr, nullptr);
}
int grid_kernel_attributes_lean(int alg, int *out) {
    if (out == nullptr || !lean_available(alg)) { g_last_error = "grid_kernel_attributes_lean: not available"; return -1; }
    hipFuncAttributes a;
    if (alg == GRID_ALG_ID_DU) G::inverse_dynamics_gradient_lean_attributes<T>(&a);
    else if (alg == GRID_ALG_FD) G::forward_dynamics_lean_attributes<T>(&a);
    else if (alg == GRID_ALG_MINV) G::direct_minv_lean_attributes<T>(&a);
    else G::forward_dynamics_gradient_lean_attributes<T>(&a);
    if (int rc = grid_check("grid_kernel_attributes_lean")) return rc;
    out[0] = a.numRegs; out[1] = (int)a.sharedSizeBytes; out[2] = (int)a.localSizeBytes; out[3] = a.maxThreadsPerBlock;
    return 0;
}
int grid_kernel_attributes_coop(int alg, int *out) {
    if (out == nullptr || !coop_available(alg)) { g_last_error = "grid_kernel_attributes_coop: not available"; return -1; }
    hipFuncAttributes a;
    G::forward_dynamics_gradient_coop_attributes<T>(&a);
    if (int rc = grid_check("grid_kernel_attributes_coop")) return rc;
    out[0] = a.numRegs; out[1] = (int)a.sharedSizeBytes; out[2] = (int)a.localSizeBytes; out[3] = a.maxThreadsPerBlock;
    return 0;
}

int grid_wave_available(int alg) { return wave_available(alg) ? 1 : 0; }
int grid_set_wave(grid_handle *h, int alg, int mode) {
    if (h == nullptr || alg < 0 || alg > 4 || mode < 0 || mode > 2) { g_last_error = "grid_set_wave: bad arguments"; return -1; }
    if (mode == 2 && !wave_available(alg)) { g_last_error = "grid_set_wave: no wave-per-configuration kernel was generated for this robot/algorithm"; return -1; }
    h->wave[alg] = mode;
    return 0;
}
int grid_get_wave(grid_handle *h, int alg, int num_timesteps) {
    if (h == nullptr || alg < 0 || alg > 4) return -1;
    return use_wave(h, alg, num_timesteps, nullptr, nullptr) ? 1 : 0;
}
int grid_kernel_attributes_wave(int alg, int *out) {
    if (out == nullptr || !wave_available(alg)) { g_last_error = "grid_kernel_attributes_wave: not available"; return -1; }
    hipFuncAttributes a;
    if (alg == GRID_ALG_FD_DU) G::forward_dynamics_gradient_wave_attributes<T>(&a);
    else G::wave_attributes<T>(alg, &a);
    if (int rc = grid_check("grid_kernel_attributes_wave")) return rc;
    out[0] = a.numRegs; out[1] = (int)a.sharedSizeBytes; out[2] = (int)a.localSizeBytes; out[3] = a.maxThreadsPerBlock;
    return 0;
}

int grid_workspace_count(int alg) { return workspace_count(alg); }
int grid_set_pipeline(grid_handle *h, int alg, int mode) {
    if (h == nullptr || alg < 0 || alg > 4 || mode < 0 || mode > 2) { g_last_error = "grid_set_pipeline: bad arguments"; return -1; }
    if (mode == 2 && workspace_count(alg) == 0) { g_last_error = "grid_set_pipeline: no two-pass variant was generated for this robot/algorithm"; return -1; }
    h->pipeline[alg] = mode;
    return 0;
}

void *grid_stream(grid_handle *h, int index) {
    if (h == nullptr || index < 0 || index >= 3) { g_last_error = "grid_stream: bad arguments"; return nullptr; }
    return (void *)h->streams[index];
}

int grid_synchronize(grid_handle *h, void *stream) {
    if (h == nullptr) { g_last_error = "grid_synchronize: NULL handle"; return -1; }
    GRID_TRY(hipSetDevice(h->device), "grid_synchronize: hipSetDevice");
    GRID_TRY(hipStreamSynchronize(pick_stream(h, stream)), "grid_synchronize");
    return 0;
}

int grid_time_device(grid_handle *h, int alg, float *d_out, const float *d_in, int stride, const float *d_qdd, const float *d_Minv,
                     int K, float gravity, int blocks, int threads, void *stream, int reps, float *ms_per_launch) {
    if (int rc = dev_prep(h, d_out, d_in, K, "grid_time_device")) return rc;
    if (reps <= 0 || ms_per_launch == nullptr) { g_last_error = "grid_time_device: bad arguments"; return -1; }
    hipStream_t s = pick_stream(h, stream);
    hipEvent_t e0, e1;
    GRID_TRY(hipEventCreate(&e0), "grid_time_device: event");
    { hipError_t ec = hipEventCreate(&e1); if (ec != hipSuccess) { (void)hipEventDestroy(e0); gpuAssert(ec, __FILE__, __LINE__); return grid_fail("grid_time_device: event"); } }
    int rc = 0;
    float ms = 0.f;
    hipError_t e = hipEventRecord(e0, s);
    for (int r = 0; r < reps && e == hipSuccess && rc == 0; r++) rc = launch_alg(h, alg, d_out, d_in, stride, d_qdd, d_Minv, K, gravity, blocks, threads, s);
    if (e == hipSuccess && rc == 0) e = hipEventRecord(e1, s);
    if (e == hipSuccess && rc == 0) e = hipEventSynchronize(e1);
    if (e == hipSuccess && rc == 0) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);      // on every exit path
    if (rc != 0) return rc;
    if (e != hipSuccess) { gpuAssert(e, __FILE__, __LINE__); return grid_fail("grid_time_device"); }
    *ms_per_launch = ms / reps;
    return 0;
}

int grid_kernel_attributes(int alg, int variant, int *out) {
    if (out == nullptr) { g_last_error = "grid_kernel_attributes: out is NULL"; return -1; }
    const void *fn = nullptr;
    typedef void (*k6)(T *, const T *, const int, const G::robotModel<T> *, const T, const int);
    typedef void (*k7)(T *, const T *, const int, const T *, const G::robotModel<T> *, const T, const int);
    typedef void (*k8)(T *, const T *, const int, const T *, const T *, const G::robotModel<T> *, const T, const int);
    typedef void (*k5)(T *, const T *, const int, const G::robotModel<T> *, const int);
    switch (alg) {
    case GRID_ALG_ID:    fn = variant ? (const void *)static_cast<k7>(&G::inverse_dynamics_kernel<T>) : (const void *)static_cast<k6>(&G::inverse_dynamics_kernel<T>); break;
    case GRID_ALG_MINV:  fn = (const void *)static_cast<k5>(&G::direct_minv_kernel<T>); break;
    case GRID_ALG_FD:    fn = (const void *)static_cast<k6>(&G::forward_dynamics_kernel<T>); break;
    case GRID_ALG_ID_DU: fn = variant ? (const void *)static_cast<k7>(&G::inverse_dynamics_gradient_kernel<T>) : (const void *)static_cast<k6>(&G::inverse_dynamics_gradient_kernel<T>); break;
    case GRID_ALG_FD_DU: fn = variant ? (const void *)static_cast<k8>(&G::forward_dynamics_gradient_kernel<T>) : (const void *)static_cast<k6>(&G::forward_dynamics_gradient_kernel<T>); break;
    default: g_last_error = "unknown algorithm id"; return -1;
    }
    hipFuncAttributes a;
    GRID_TRY(hipFuncGetAttributes(&a, fn), "grid_kernel_attributes");
    out[0] = a.numRegs; out[1] = (int)a.sharedSizeBytes; out[2] = (int)a.localSizeBytes; out[3] = a.maxThreadsPerBlock;
    return 0;
}

int grid_kernel_attributes_split(int alg, int split, int *out) {
    if (out == nullptr) { g_last_error = "grid_kernel_attributes_split: out is NULL"; return -1; }
    if (split <= 1) return grid_kernel_attributes(alg, 0, out);
    hipFuncAttributes a;
    bool ok = false;
    if (alg == GRID_ALG_FD_DU) ok = G::forward_dynamics_gradient_split_attributes<T>(split, &a);
    else if (alg == GRID_ALG_ID_DU) ok = G::inverse_dynamics_gradient_split_attributes<T>(split, &a);
    if (!ok) { g_last_error = "grid_kernel_attributes_split: this split was not generated for this robot/algorithm"; return -1; }
    if (int rc = grid_check("grid_kernel_attributes_split")) return rc;
    out[0] = a.numRegs; out[1] = (int)a.sharedSizeBytes; out[2] = (int)a.localSizeBytes; out[3] = a.maxThreadsPerBlock;
    return 0;
}

}  // extern "C"
