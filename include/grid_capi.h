/*
 * grid_capi.h -- C ABI of the MI355X-native GRiD library (one shared object per robot model).
 *
 * The reference (robot-acceleration/GRiDCodeGenerator) has NO C ABI: its product is a CUDA header of
 * C++ templates that the user's own nvcc main() includes.  This shim is the
 * FFI a non-C++ host (Python ctypes here) binds instead; every entry point names the emitted
 * reference interface it stands for.  Plain pointers and sizes only; all functions return 0 on
 * success or a hipError_t-style non-zero code (never exit()), with text from grid_last_error().
 *
 * Buffer layouts are the reference's gridData<T> layouts, T = float (GRiDCodeGenerator.py:92-137):
 *   q_qd_u[K][3n] = [q | qd | u]    q_qd[K][2n]    q[K][n]    qdd[K][n]
 *   c[K][n]   Minv[K][n*n] column-major, upper triangle (lower half 0)   qdd[K][n]
 *   dc_du[K][2*n*n], df_du[K][2*n*n]: n x 2n column-major = [d/dq | d/dqd]
 */
#ifndef GRID_CAPI_H
#define GRID_CAPI_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grid_handle grid_handle;

/* algorithm ids (order of the reference's <FUNC_CODE> list, GRiDCodeGenerator.py:279) */
enum { GRID_ALG_ID = 0, GRID_ALG_MINV = 1, GRID_ALG_FD = 2, GRID_ALG_ID_DU = 3, GRID_ALG_FD_DU = 4 };

/* ---- model / build constants (reference: the `const int` block, GRiDCodeGenerator.py:75-83) ---- */
const char *grid_robot_name(void);
int grid_num_joints(void);                                   /* NUM_JOINTS */
/* out[0..9] = NUM_JOINTS, ID/MINV/FD/ID_DU/FD_DU_DYNAMIC_SHARED_MEM_COUNT, ID_DU/FD_DU_MAX_SHARED_MEM_COUNT,
 *             SUGGESTED_THREADS, SUGGESTED_MAX_BLOCKS */
int grid_constants(int *out, int count);
const char *grid_compute_dtype(void);                        /* "f32" or "f64": arithmetic type for T=float */
const char *grid_last_error(void);

/* ---- lifecycle ---- */
/* init_robotModel<T>() + init_grid<T>() on `device` (GRiDCodeGenerator.py:155-189, helpers/_topology_helpers.py:365-380) */
int grid_init(int device, grid_handle **out);
/* init_gridData<T>(max_timesteps) (GRiDCodeGenerator.py:116-153); needed only by the host-buffer calls */
int grid_alloc(grid_handle *h, int max_timesteps);
/* close_grid<T>(streams, d_robotModel, hd_data) (GRiDCodeGenerator.py:191-203) */
int grid_close(grid_handle *h);
/* copy back what init_robotModel uploaded: h_XImats[72n], h_topology[grid_topology_helpers_count()] */
int grid_topology_helpers_count(void);
int grid_read_model(grid_handle *h, float *h_XImats, int *h_topology);

/* ---- host-buffer calls = the reference's host wrappers, mode 0 (H2D, kernel, D2H, synchronous) ----
 * inverse_dynamics<T,USE_QDD_FLAG>            algorithms/_inverse_dynamics.py:423-495   (h_qdd may be NULL)
 * direct_minv<T>                              algorithms/_direct_minv.py:456-517
 * forward_dynamics<T>                         algorithms/_forward_dynamics.py:196-252
 * inverse_dynamics_gradient<T,USE_QDD_FLAG>   algorithms/_inverse_dynamics_gradient.py:762-834
 * forward_dynamics_gradient<T,USE_QDD_MINV>   algorithms/_forward_dynamics_gradient.py:179-242 (h_qdd and h_Minv both NULL or both set)
 */
int grid_inverse_dynamics(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, float *h_c, int num_timesteps, float gravity);
int grid_direct_minv(grid_handle *h, const float *h_q_qd_u, float *h_Minv, int num_timesteps);
int grid_forward_dynamics(grid_handle *h, const float *h_q_qd_u, float *h_qdd, int num_timesteps, float gravity);
int grid_inverse_dynamics_gradient(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, float *h_dc_du, int num_timesteps, float gravity);
int grid_forward_dynamics_gradient(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, const float *h_Minv, float *h_df_du,
                                   int num_timesteps, float gravity);

/* ---- device-pointer calls = the kernels themselves (reference mode 2, `_compute_only`), asynchronous ----
 * Arguments mirror the __global__ signatures: _inverse_dynamics.py:365-369, _direct_minv.py:414, _forward_dynamics.py:151-152,
 * _inverse_dynamics_gradient.py:703-707, _forward_dynamics_gradient.py:117-125.
 * blocks/threads <= 0 selects the suggested launch shape.
 * stream == NULL is HIP's default stream, i.e. the launch is ORDERED with whatever the caller queued before it on the default
 * stream (hipMemcpy, a torch fill, ...): no synchronisation is needed between producing the inputs there and launching.  Any
 * other value is a hipStream_t.  The handle's own streams -- created by init_grid exactly as the reference creates them, with
 * priorities and hipStreamNonBlocking (GRiDCodeGenerator.py:182-188), hence NOT ordered with the default stream -- are used by
 * the host-buffer calls above and can be had from grid_stream(h, 0..2) by a caller that orders its own work on them.
 */
int grid_inverse_dynamics_device(grid_handle *h, float *d_c, const float *d_q_qd, int stride_q_qd, const float *d_qdd,
                                 int num_timesteps, float gravity, int blocks, int threads, void *stream);
int grid_direct_minv_device(grid_handle *h, float *d_Minv, const float *d_q, int stride_q,
                            int num_timesteps, int blocks, int threads, void *stream);
int grid_forward_dynamics_device(grid_handle *h, float *d_qdd, const float *d_q_qd_u, int stride_q_qd_u,
                                 int num_timesteps, float gravity, int blocks, int threads, void *stream);
int grid_inverse_dynamics_gradient_device(grid_handle *h, float *d_dc_du, const float *d_q_qd, int stride_q_qd, const float *d_qdd,
                                          int num_timesteps, float gravity, int blocks, int threads, void *stream);
int grid_forward_dynamics_gradient_device(grid_handle *h, float *d_df_du, const float *d_q_qd_u, int stride_q_qd_u,
                                          const float *d_qdd, const float *d_Minv,
                                          int num_timesteps, float gravity, int blocks, int threads, void *stream);
int grid_synchronize(grid_handle *h, void *stream);
/* the handle's own non-blocking streams (init_grid<T>(): three, descending priority); NULL for a bad index */
void *grid_stream(grid_handle *h, int index);

/* ---- rollout consumer of the forward-dynamics gradient (no reference kernel; the reference ships the `_device` tier for
 * exactly this use, README.md:26-29, algorithms/_forward_dynamics_gradient.py:59-99) ----
 * One lane integrates one trajectory for num_steps semi-implicit Euler steps (qdd = FD(q, qd, u_t); qd += dt qdd; q += dt qd)
 * with q, qd kept in registers, and writes per step the next state and the discrete linearisation:
 *   d_traj[(t*num_timesteps + k)*grid_rollout_row_count() + ...] = [x_{t+1} (2n) | A_t (2n x 2n col-major) | B_t (2n x n col-major)]
 *   d_x0[k][2n] = [q | qd];  d_u_traj[(t*num_timesteps + k)*n + j]   (both time-major, dense).  Asynchronous on `stream`. */
int grid_rollout_row_count(void);
int grid_forward_dynamics_gradient_rollout_device(grid_handle *h, float *d_traj, const float *d_x0, const float *d_u_traj,
                                                  int num_timesteps, int num_steps, float dt, float gravity,
                                                  int blocks, int threads, void *stream);

/* ---- column-split variants of the two gradient kernels (no reference counterpart) ----
 * For small batches the chip is under-filled with one lane per configuration; the generator therefore also emits
 * kernels in which S blocks share a tile and each computes a group of gradient columns (repeating the common prefix).
 * grid_splits: the S values generated for `alg` (returns their number).  grid_set_split: 0 = automatic (default),
 * 1 = never, S = force.  grid_get_split: the S a call with `num_timesteps` would use.  Results are bit-identical. */
int grid_splits(int alg, int *out, int count);
int grid_set_split(grid_handle *h, int alg, int split);
int grid_get_split(grid_handle *h, int alg, int num_timesteps);

/* ---- tile-cooperative forward-dynamics-gradient kernel (no reference counterpart; the regime it serves is the reference's own:
 * one thread block per configuration, GRiDCodeGenerator.py:72-83, algorithms/_forward_dynamics_gradient.py:7-57) ----
 * One block of 4 wavefronts per tile of 64 configurations: one wave runs the Minv recursion while the others run RNEA, the
 * results cross through LDS, then every wave differentiates its own group of columns.  Unlike the column-split kernels the
 * shared prefix is computed ONCE per tile.  grid_coop_available: 1 if the generator emitted it for `alg` (GRID_ALG_FD_DU).
 * grid_set_coop: 0 = automatic (default: FD_DU_COOP_AUTO_MIN_TILES of the generated header -- every batch size for large robots in
 * fp32, from 192 tiles on in the mixed arithmetic, never for small robots), 1 = never, 2 = always.
 * grid_get_coop: which tile-cooperative kernel a call with `num_timesteps` would dispatch -- 0 none, 1 the 4-wave kernel, 2 its
 * register-lean 8-wave variant (it takes precedence over the column split).
 * Register-lean variant (`forward_dynamics_gradient_kernel_coop8`, large robots, fp32 arithmetic): EIGHT wavefronts per tile, two per
 * SIMD at <= 256 registers each -- a block-shared input table, the Minv recursion by base-rooted tree with its carried values parked in
 * LDS, gradient HALF columns as work items -- so that a second wave fills the issue slots and waits a lone 464-register wave leaves
 * empty.  grid_lean_available: 1 if emitted; grid_set_coop mode 3 = always this variant; automatic from FD_DU_LEAN_AUTO_MIN_TILES of
 * the generated header (0: on request only).  Its output leaves in contiguous runs per wave, cut at the 32-byte sectors of the row.
 * The same block serves the INVERSE-dynamics gradient at qdd = 0 (`inverse_dynamics_gradient_kernel_coop8`, alg = GRID_ALG_ID_DU:
 * input table, one barrier, gradient half-columns; replaces algorithms/_inverse_dynamics_gradient.py:199-246,501-540's
 * block-per-configuration mapping for large robots): grid_lean_available(GRID_ALG_ID_DU), grid_set_coop(h, GRID_ALG_ID_DU, 0|1|3),
 * automatic from ID_DU_LEAN_AUTO_MIN_TILES on; a call with d_qdd != NULL keeps the lane-per-configuration kernel.
 * And FORWARD DYNAMICS itself (`forward_dynamics_kernel_coop8`, alg = GRID_ALG_FD: the gradient kernel's prefix -- input table, shared
 * Minv recursion, bias torques, qdd rows -- then one wave writes qdd; replaces algorithms/_forward_dynamics.py:21-112's block per
 * configuration for large robots): grid_lean_available(GRID_ALG_FD), grid_set_coop(h, GRID_ALG_FD, 0|1|3), FD_LEAN_AUTO_MIN_TILES.
 * And the direct Minv (`direct_minv_kernel_coop8`, alg = GRID_ALG_MINV: input table, backward pass per tree, forward pass over all waves,
 * columns written from registers; reads q only; replaces algorithms/_direct_minv.py:23-382's block per configuration): the same calls
 * with GRID_ALG_MINV, MINV_LEAN_AUTO_MIN_TILES / _MAX_TILES. */
int grid_coop_available(int alg);
int grid_lean_available(int alg);
int grid_set_coop(grid_handle *h, int alg, int mode);
int grid_get_coop(grid_handle *h, int alg, int num_timesteps);
int grid_kernel_attributes_coop(int alg, int *out);
int grid_kernel_attributes_lean(int alg, int *out);

/* ---- wave-per-configuration kernels (all five algorithms): the small-batch path ----
 * The reference's own mapping -- one thread block per configuration whose threads split the 6x6 products and the gradient columns
 * (GRiDCodeGenerator.py:72-83 SUGGESTED_THREADS, helpers/_code_generation_helpers.py:41-55, algorithms/_inverse_dynamics_gradient.py:
 * 199-246,501-540) -- redone for 64-wide wavefronts: one block per configuration, one wavefront per group of base-rooted trees, lane =
 * gradient column / Minv column / joint; column-independent quantities are wave-uniform (v_readlane broadcasts), Minv crosses lanes
 * through wave-local LDS.  A batch of K <= #CUs configurations then costs one configuration's chain on 64 lanes instead of the
 * whole chain on one lane (Atlas-30, K = 64, us per launch, lanes -> waves: RNEA 9.6 -> 5.2, Minv 31 -> 7.6, FD 28 -> 10.1,
 * RNEA gradient 37 -> 12.7, FD gradient 55 -> 19.9).  grid_wave_available: 1 if emitted for `alg` (every algorithm of a robot that
 * has the kernels; none in all-double builds).  grid_set_wave: 0 = automatic (default: batches up to <ALG>_WAVE_AUTO_MAX_K of the
 * generated header, which quotes the measurements behind each constant, capped at <ALG>_LEAN_WAVE_MAX_K where the library has a
 * register-lean tile-cooperative kernel that is faster from a smaller batch on -- and only for calls that leave the launch shape to the
 * library, blocks <= 0 and threads <= 0: a caller's blocks x threads means blocks of `threads` CONFIGURATIONS, the reference's launch
 * shape, and keeps the lane-per-configuration kernels), 1 = never, 2 = always (`blocks` is then the number of configurations in
 * flight, one block each).  grid_get_wave: 1 if a call with `num_timesteps` and the default launch shape would dispatch it (it takes
 * precedence over the other variants).  The wave kernels take the optional d_qdd of
 * RNEA and its gradient; a forward-dynamics-gradient call with precomputed d_qdd/d_Minv keeps to the lane-per-configuration kernel. */
int grid_wave_available(int alg);
int grid_set_wave(grid_handle *h, int alg, int mode);
int grid_get_wave(grid_handle *h, int alg, int num_timesteps);
int grid_kernel_attributes_wave(int alg, int *out);

/* ---- two-pass (workspace) variants of the gradient kernels (no reference counterpart) ----
 * For robots whose gradient working set exceeds the register file (Atlas-30) the generator also emits a two-kernel
 * variant: pass 1 (RNEA [+ Minv, qdd]) writes per-joint quantities to a tile-major SoA workspace in HBM, pass 2 runs the
 * gradient column by column re-reading them.  grid_workspace_count: elements per configuration (0 = not generated).
 * grid_set_pipeline: 0 = automatic (single kernel: measured faster since the recomputing column schedule), 1 = single kernel,
 * 2 = two-pass (error when the generator emitted none for this robot).
 * The workspace lives in the handle and grows on demand (first call at a new batch size allocates).  ONE workspace serves
 * both gradient algorithms: a two-pass launch on a stream other than the one that used it last first waits for that
 * stream (hipStreamSynchronize), and growth waits for the whole device -- neither is graph-capturable; the single-kernel
 * default has no such state. */
int grid_workspace_count(int alg);
int grid_set_pipeline(grid_handle *h, int alg, int mode);

/* ---- measurement ----
 * `reps` back-to-back launches of algorithm `alg` on `stream`, bracketed by hipEvents recorded on that same stream;
 * *ms_per_launch = elapsed / reps.  (Replaces the reference's `_single_timing` clock_gettime twins,
 * algorithms/_forward_dynamics_gradient.py:229-241, which time block latency rather than throughput.) */
int grid_time_device(grid_handle *h, int alg, float *d_out, const float *d_in, int stride, const float *d_qdd, const float *d_Minv,
                     int num_timesteps, float gravity, int blocks, int threads, void *stream, int reps, float *ms_per_launch);
/* hipFuncGetAttributes of the kernel for `alg` (variant: 0 default inputs, 1 with qdd / qdd+Minv):
 * out[0]=numRegs out[1]=static LDS bytes out[2]=scratch (local) bytes per lane out[3]=maxThreadsPerBlock */
int grid_kernel_attributes(int alg, int variant, int *out);
/* the same for the S-way column-split kernel of a gradient algorithm (the kernel grid_get_split() says a call dispatches);
 * split <= 1 is the unsplit kernel; an S that was not generated is an error */
int grid_kernel_attributes_split(int alg, int split, int *out);

#ifdef __cplusplus
}
#endif
#endif /* GRID_CAPI_H */
