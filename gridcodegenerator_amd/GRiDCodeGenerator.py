"""Generator facade: ``GRiDCodeGenerator(robot).gen_all_code()`` writes ``<FILE_NAMESPACE>.hip.h``.

Drop-in boundary (reference GRiDCodeGenerator.py:37 and :241): same constructor arguments, same
``gen_all_code(use_thread_group=False, include_base_inertia=False)`` entry point, output written to the
current working directory and left in ``.code_str``; the emitted header exposes the same
``ALGORITHM_inner / _device / _kernel / host`` names and buffer layouts.  What is emitted is HIP for
gfx950 with one wavefront lane per configuration, not the reference's
one-block-per-configuration CUDA.

Generation-time knobs that have no reference counterpart are keyword-only:
    precision        "fp32" | "mixed" | "fp64"  arithmetic of the kernels for T=float (I/O stays T).  "fp32": compute type
                                      C = float.  "mixed": C = float, but the Minv recursion and qdd = Minv (u - c) -- the parts
                                      whose round-off cond(M) amplifies -- run in double.  "fp64": C = double everywhere (verified on
                                      the GPU as a regression variant, 4.7x the time of fp32 for Atlas-30: DESIGN.md section 4)
    trig             "fast" | "libm" | "f64"   inline float sincos (default), library sincosf, or double then rounded
                                      as the reference does (helpers/_topology_helpers.py:127-128)
    suggested_threads, max_threads    threads per block the LDS counts are sized for (multiple of 64) / __launch_bounds__
    suggested_max_blocks              grid size cap of the host wrappers (larger batches grid-stride)
    out_chunk                         max values per configuration staged in LDS per coalesced flush
    emit_inner_api                    also emit the pointer-style ``_inner`` tier (API parity)
    pipeline         "auto" | bool    also emit the two-pass (workspace) variants of the gradient kernels; auto: n > 12
    grad_schedule    "auto" | "fused" | "recompute"   body of the single-kernel gradient cores: demand-ordered fused trace, or
                     column-serial with per-column recomputation of v, a, f (large robots); auto: recompute for n > 12.
                     ("fused" with n > 12 spills kilobytes per lane; correct since the kernels are branch-free)
    grad_splits      "auto" | list    column-split variants of the two gradient kernels to emit (small-batch speed)
    waves_per_simd                    __launch_bounds__ occupancy hint for the unsplit kernels (caps registers at 512/w)
    prismatic_gradient "corrected" | "reference"   the d/dq seed of a prismatic joint's own column: force cross product (agrees with
                                      finite differences; default) or the reference's motion cross product (_test.py:311,437) so that
                                      such a robot's gradients can be compared with the reference's numbers; identical for revolute joints
    allow_unverified bool             accept what reintroduces lane-divergent control flow (trig="libm"/"f64", out_mode="direct"):
                                      such kernels are unverified on the GPU
    experimental     dict             measured-and-rejected or test-only variants, NOT part of the supported surface
                                      (defaults in EXPERIMENTAL_DEFAULTS; each is described and its measurement quoted there)
"""
from .algorithms._emit import AlgorithmEmitMixin
from .emit.model import RobotSpec
from .helpers._runtime_emit import RuntimeEmitMixin
from .helpers._spatial_emit import SpatialAlgebraEmitMixin
from .helpers._text import TextMixin
from .verification import VerificationMixin


class GridGenerationWarning(UserWarning):
    """A kernel family was left out, or a kernel is predicted to exceed the register file, for this robot (see `generation_notes`)."""


class GRiDCodeGenerator(TextMixin, RuntimeEmitMixin, SpatialAlgebraEmitMixin, AlgorithmEmitMixin, VerificationMixin):
    # Experiments that were built, measured on MI355X and rejected (or that exist for the tests only).  Kept reachable through
    # ``experimental={...}`` so the measurements in DESIGN.md stay reproducible; none of them is a supported option.
    EXPERIMENTAL_DEFAULTS = dict(
        emit_order="demand",    # "creation": nodes in trace order (the recomputing cores always use it)
        fence_every=0,          # extra scheduling fence every N statements (a fence always follows each output store: without it
                                # hipcc hoists every output's dot product above the stores, iiwa-7 dFD 472 vs 257 registers)
        out_mode="staged",      # "direct": per-lane row stores instead of LDS-staged flat stores (15.0 vs 12.5 us at K=16384)
        packed=False,           # (d/dq, d/dqd) recursions as v_pk_fma_f32 pairs: 6777 -> 4818 instructions but 512 registers + spills
        grad_table=False,       # recompute schedule: sin q, cos q, qd, qdd in a per-wave LDS table (Atlas-30: 186 vs 163 us); mixed5 tests it
        split_fences=True,      # scheduling fence after every output store also in the S >= 3 split kernels (without: 13.0 vs 12.0 us)
        split_sets=True,        # column groups of the S >= 3 splits of small robots as arbitrary column SETS (exact exhaustive partition on
                                # the traced costs; iiwa-7 dFD x4: heaviest group 3580 -> 2937 operations), flushed once per half
                                # (grid_out_colset).  Round 2 flushed them per column and measured them slower at K = 16384 (12.05 vs
                                # 11.46 us); with one flush per half and a tile's groups in one block: 9.9 vs 10.6 us (profiles/r03)
        fence_stride=1,         # fence after every N-th store instead of every store (12-15 % slower, spills)
        dot_ways=1,             # dot products over w interleaved accumulators (0-4 % slower)
        split_cap=(2, 3, 4),    # which split factors are compiled for two waves per SIMD (<= 256 registers; small robots only).  The 3- and
                                # 4-way splits run one wave per SIMD at the batch sizes they serve; the cap costs them 2-6 spilled values
                                # (9.94 vs 9.86 us at K = 16384) and lets a second stream's launch share the chip.  Capping the 7-way split
                                # so that its waves PAIR UP on SIMDs loses: 14.8 vs 11.1 us (profiles/r03/two_waves_per_simd.md)
        split_half_columns=True,  # ... and the sets may hold HALF columns (d/dq and d/dqd of a column in different groups) where that
                                # lowers the heaviest group: iiwa-7 dFD x4 2937 -> 2771 operations, dID x4 1658 -> 1507
        in_rows=False,          # inputs by per-lane 16-byte row loads instead of coalesced loads staged through LDS (grid_rows)
        split_flush_slots=10,   # half-column sets of the S >= 3 splits balanced on arithmetic + this many issue slots per OUTPUT value (the stamped
                                # 4-way kernel: ~43 cycles = ~10 slots per value).  iiwa-7 dFD K = 16384: 9.32-9.52 us (0) -> 8.83-8.96 (10), 9.09 (20)
                                # -- profiles/r04/exp_iiwa7_asymmetric_pairs.txt
        split_asym=0,           # > 0: also emit the ASYMMETRIC 8-way split (4 heavy groups on the waves dispatched first, 4 light ones whose
                                # cost counts this many times) as `*_kernel_split8`, one 512-thread block per tile.  Measured 9.31-9.37 us at
                                # 1.4-2.5 against 8.83-8.96 for the flush-balanced 4-way split: rejected (same file)
        wave_occupancy=1,       # wave-per-configuration kernels: waves per SIMD the register allocation must leave room for (2: <= 256 registers)
        lean_read_ahead=0,      # register-lean 8-wave kernel: LDS reads issued this many instructions ahead of their use (Tracer.emit read_ahead)
        lean_loop_per_role=False,   # register-lean inverse-dynamics-gradient kernel: one tile loop per role (switch outside) instead of one switch per tile
        lean_mixed=False,       # mixed arithmetic: also build the register-lean forward-dynamics-gradient kernel (see algorithms/_emit.py: _lean_prepare)
        lean_min_joints=13,     # register-lean 8-wave kernels for robots with at least this many joints
        lean_id_plan={},        # register-lean inverse-dynamics-gradient kernel: keyword overrides of cores.lean_plan_id (chain_f)
        lean_plan={},           # register-lean 8-wave kernel: keyword overrides of cores.lean_plan (younger_speed, max_parked)
        lean_probe=None,        # register-lean 8-wave kernel, timing probes (NOT a correct kernel): "prefix" = phases 0-2 only, "older" / "younger" =
                                # only the waves dispatched first / last keep their gradient half-columns (profiles/r04/lean_probes.txt)
        coop_hoist=False,       # tile-cooperative cores of small robots: force everything that does not depend on qdd in front of
                                # the first barrier (and let the producer go without columns): 12.9 vs 12.1 us at K=16384
    )

    def __init__(self, robotObj, DEBUG_MODE=False, NEED_PRINT_MAT=False, USE_DYNAMIC_SHARED_MEM=True,
                 FILE_NAMESPACE="grid", *, precision="fp32", trig="fast", suggested_threads=64, max_threads=256,
                 suggested_max_blocks=2048, out_chunk=64, emit_inner_api=True, pipeline="auto", grad_schedule="auto",
                 grad_splits="auto", waves_per_simd=1, allow_unverified=False, experimental=None, prismatic_gradient="corrected"):
        if precision not in ("fp32", "mixed", "fp64"):
            raise ValueError("precision must be 'fp32', 'mixed' or 'fp64'")
        if trig not in ("fast", "libm", "f64"):
            raise ValueError("trig must be 'fast', 'libm' or 'f64'")
        if suggested_threads % 64 != 0 or not (64 <= suggested_threads <= max_threads <= 1024):
            raise ValueError("need 64 <= suggested_threads <= max_threads <= 1024, suggested_threads a multiple of 64")
        if grad_schedule not in ("auto", "fused", "recompute"):
            raise ValueError("grad_schedule must be 'auto', 'fused' or 'recompute'")
        if prismatic_gradient not in ("corrected", "reference"):
            raise ValueError("prismatic_gradient must be 'corrected' (force cross product: agrees with finite differences) or 'reference' "
                             "(the reference's motion cross product, _test.py:311,437: identical for revolute joints)")
        self.prismatic_gradient = prismatic_gradient
        exp = dict(self.EXPERIMENTAL_DEFAULTS)
        unknown = set(experimental or {}) - set(exp)
        if unknown:
            raise ValueError("unknown experimental option(s): %s" % sorted(unknown))
        exp.update(experimental or {})
        if exp["out_mode"] not in ("direct", "staged"):
            raise ValueError("out_mode must be 'direct' or 'staged'")
        self.robot = robotObj
        self.spec = RobotSpec(robotObj)
        large = self.spec.n > 12
        # Round 1's failing variants (fp64, fused n > 12, register caps on large robots: DESIGN.md section 9) were spill code inside
        # reduced-EXEC regions (section 9.1); the kernels are branch-free now and the variants are regression-tested on the GPU
        # (tests/test_round1_regressions.py: all three pass, and so does an all-double Atlas-30 library with 4 KB of scratch per
        # lane).  What is still refused is what reintroduces lane-divergent control flow.
        if not allow_unverified:
            if trig != "fast":
                raise ValueError("trig=%r inlines the math library's sincos, whose large-argument path is a lane-divergent branch: kernels "
                                 "with such branches are unverified (hipcc placed spill code inside the masked region, DESIGN.md "
                                 "section 9); use trig='fast' (branch-free, |q| <= 1e6) or pass allow_unverified=True" % trig)
            if exp["out_mode"] == "direct":
                raise ValueError("out_mode='direct' wraps the core in a per-lane `if`: lane-divergent control flow is unverified "
                                 "(DESIGN.md section 9); pass allow_unverified=True to build it anyway")
        self.allow_unverified = bool(allow_unverified)
        self._chunks = []
        self.indent_level = 0
        self.DEBUG_MODE = DEBUG_MODE
        self.gen_print_mat = DEBUG_MODE or NEED_PRINT_MAT
        # kept for signature parity: the lane-per-configuration kernels always use dynamic LDS
        self.use_dynamic_shared_mem_flag = True
        self.file_namespace = FILE_NAMESPACE
        self.precision = precision
        self.trig = trig
        self.suggested_threads = int(suggested_threads)
        self.max_threads = int(max_threads)
        self.suggested_max_blocks = int(suggested_max_blocks)
        self.out_chunk = int(out_chunk)
        self.grad_splits = grad_splits
        self.waves_per_simd = int(waves_per_simd)
        self.use_pipeline = large if pipeline == "auto" else bool(pipeline)
        self.grad_schedule = ("recompute" if large else "fused") if grad_schedule == "auto" else grad_schedule
        self.emit_order = exp["emit_order"]
        self.fence_every = int(exp["fence_every"])
        self.out_mode = exp["out_mode"]
        self.packed = bool(exp["packed"])
        self.grad_table = bool(exp["grad_table"]) and self.grad_schedule == "recompute"
        self.split_fences = bool(exp["split_fences"])
        self.split_sets = bool(exp["split_sets"])
        self.fence_stride = int(exp["fence_stride"])
        self.dot_ways = int(exp["dot_ways"])
        self.split_cap = tuple(int(x) for x in exp["split_cap"])
        self.coop_hoist = bool(exp["coop_hoist"])
        self.lean_probe = exp["lean_probe"]
        self.wave_occupancy = int(exp["wave_occupancy"])
        self.split_flush_slots = exp["split_flush_slots"] if exp["split_flush_slots"] == "flush" else float(exp["split_flush_slots"])
        self.split_asym = float(exp["split_asym"])
        self.lean_plan_options = dict(exp["lean_plan"])
        self.lean_id_plan_options = dict(exp["lean_id_plan"])
        self.lean_read_ahead = int(exp["lean_read_ahead"])
        self.lean_min_joints = int(exp["lean_min_joints"])
        self.lean_mixed = bool(exp["lean_mixed"])
        self.lean_loop_per_role = bool(exp["lean_loop_per_role"])
        self.in_rows = bool(exp["in_rows"])
        self.split_half_columns = bool(exp["split_half_columns"])
        self.lean_wave_max_k, self.lean_id_wave_max_k = 768, 1408     # batch sizes from which the lean kernels beat the wave-per-configuration kernels
        self.lean_minv_auto = (1, 256, 1152)      # (profiles/r04/lean_minv_sweep.txt)      # register-lean direct-Minv kernel: automatic from / up to this many tiles; wave-per-configuration kernel up to this batch size
        self.lean_fd_auto_min_tiles, self.lean_fd_auto_max_tiles, self.lean_fd_wave_max_k = 1, 512, 640      # register-lean forward-dynamics kernel (profiles/r04/lean_fd_sweep.txt)
        self.lean_id_auto_min_tiles = 1    # register-lean inverse-dynamics-gradient kernel: automatic from this many tiles on
        self.lean_auto_min_tiles = 1       # register-lean 8-wave kernel: automatic from this many tiles on (0: on request) -- measured faster than
                                           # the 4-wave kernel at every batch size (profiles/r04/lean_sweep.txt)
        self.wave_auto_max_k = 1024        # batch sizes up to which large robots use the wave-per-configuration kernel by themselves
        # ... for the other algorithms, by robot size: measured on MI355X with tools/latency_all.py (profiles/r03/latency_all_*.txt), us per
        # launch lane-per-configuration / wave-per-configuration.  Atlas-30: ID 9.9 / 6.6 at K = 1024 (9.9 / 13.9 at 2048), MINV 32.4 / 27.8
        # at 2048 (33 / 53 at 4096), FD 28.7 / 19.6 at 1024 (28.8 / 37.8 at 2048), ID_DU 40.4 / 34.8 at 2048 (45 / 62 at 4096).  iiwa-7:
        # ID 3.6 / 3.2 at 512, MINV 4.8 / 3.4 at 1024, FD 5.0 / 4.5 at 512, gradients slower at every K.  mixed-5: no gain.
        self.wave_auto_other = ({"ID": 1024, "MINV": 2048, "FD": 1024, "ID_DU": 2048} if self.spec.n > 12 else
                                {"ID": 512, "MINV": 1024, "FD": 512, "ID_DU": 0} if self.spec.n >= 7 else {"ID": 0, "MINV": 0, "FD": 0, "ID_DU": 0})
        self.kernel_instances = []
        self.split_stats = {}
        self.generation_notes = []          # what was left out / is predicted to spill for this robot (GridGenerationWarning)
        self.predicted_live = {}
        self.emit_inner_api = bool(emit_inner_api)
        self.core_stats = {}
        self.trace_stats = {}
        self._build_io_layout()

    # reference-compatible integer bookkeeping (helpers/_topology_helpers.py:184-215)
    def gen_topology_helpers_size(self):
        return self.spec.topology_helpers_size()

    def gen_topology_sparsity_helpers_python(self, INIT_MODE=False):
        t = self.spec.sparsity_tables()
        if INIT_MODE:
            return ([str(v) for v in t["num_ancestors"]], [str(v) for v in t["num_subtree"]],
                    [str(v) for v in t["running_sum_num_ancestors"]], [str(v) for v in t["running_sum_num_subtree"]])
        return (t["dva_cols_per_partial"], t["dva_cols_per_jid"], t["running_sum_dva_cols_per_jid"],
                t["df_cols_per_partial"], t["df_cols_per_jid"], t["running_sum_df_cols_per_jid"], t["df_col_that_is_jid"])

    def gen_kernel_instance_list(self):
        """Macros that let a build instantiate ONE kernel per translation unit (parallel compiles):
        GRID_KERNEL_INST_<k>(KW) expands to `KW <kernel specialisation k for T = float>;` with KW = `template` or
        `extern template`.  Requires `typedef float T;` at the point of use."""
        ns = self.file_namespace
        self.gen_add_code_line("")
        self.gen_add_code_line("// ---- explicit-instantiation list (T = float) for one-kernel-per-translation-unit builds ----")
        self.gen_add_code_line("#define GRID_NUM_KERNEL_INSTANCES %d" % len(self.kernel_instances))
        for k, decl in enumerate(self.kernel_instances):
            self.gen_add_code_line("#define GRID_KERNEL_INST_%d(KW) KW %s" % (k, decl.replace("@NS", ns)))
        self.gen_add_code_line("#define GRID_FOR_EACH_KERNEL_INST(KW) " + " ".join("GRID_KERNEL_INST_%d(KW)" % k
                                                                                   for k in range(len(self.kernel_instances))))

    def note(self, text):
        """Record something a user of this robot's header should know (a kernel family that was not emitted, a predicted spill)."""
        self.generation_notes.append(text)

    def _report_generation_notes(self):
        """The reference emits for any robot object and lets nvcc / the GPU find out (GRiDCodeGenerator.py:37-46: dynamic shared memory
        forced above 12 joints; helpers/_topology_helpers.py:193-258: tables for any tree).  Here the straight-line design has limits
        that are known at generation time -- 2m <= 64 lanes per wave group, an exchange region that must fit 160 KB of LDS, 512
        registers per lane -- so the generator SAYS what it left out or expects to spill instead of leaving it to the build."""
        import warnings
        n = self.spec.n
        if n > 32 and self.grad_schedule == "recompute":
            from .emit import cores
            tr = cores.core_gradient_recompute(self.spec, "fd")
            mx, _ = tr.max_live()
            self.predicted_live["forward_dynamics_gradient_kernel"] = mx
            if mx > 480:
                self.note("forward_dynamics_gradient_kernel keeps %d values alive per lane (the register file holds 512): expect it to spill "
                          "to scratch; host.build_library refuses a kernel beyond GRID_MAX_SCRATCH = %s B per lane" % (mx, "8192"))
        if self.generation_notes:
            warnings.warn(GridGenerationWarning("%s (%d joints): %s" % (self.spec.name, n, "; ".join(self.generation_notes))), stacklevel=3)

    def output_file_name(self):
        return self.file_namespace + ".hip.h"

    def gen_all_code(self, use_thread_group=False, include_base_inertia=False):
        if use_thread_group:
            raise NotImplementedError("use_thread_group (cooperative groups) is an unfinished placeholder in the reference "
                                      "(algorithms/_inverse_dynamics.py:391) and has no place in a lane-per-configuration design")
        if include_base_inertia:
            raise NotImplementedError("include_base_inertia adds data no reference emitter reads (helpers/_topology_helpers.py:5-12)")
        from .emit.trace import Tracer
        from .emit import algorithms as alg_mod
        saved = (Tracer.use_packed, Tracer.mixed, Tracer.dot_ways, alg_mod.PRISMATIC_GRADIENT)
        Tracer.use_packed = self.packed
        Tracer.mixed = (self.precision == "mixed")
        Tracer.dot_ways = self.dot_ways
        alg_mod.PRISMATIC_GRADIENT = self.prismatic_gradient
        try:
            self._gen_all_code_body(use_thread_group, include_base_inertia)
        finally:
            Tracer.use_packed, Tracer.mixed, Tracer.dot_ways, alg_mod.PRISMATIC_GRADIENT = saved

    def _gen_all_code_body(self, use_thread_group, include_base_inertia):
        self._chunks = []
        self.indent_level = 0
        self.core_stats = {}
        self.trace_stats = {}
        self.kernel_instances = []
        self.split_stats = {}
        self.generation_notes = []
        self.predicted_live = {}
        if hasattr(self, "_lean_cache"):
            del self._lean_cache
        n = self.spec.n
        file_notes = [
            "Interface is:",
            "    __host__   robotModel<T> *d_robotModel = init_robotModel<T>()",
            "    __host__   hipStream_t *streams = init_grid<T>()",
            "    __host__   gridData<T> *hd_data = init_gridData<T,NUM_TIMESTEPS>();",
            "    __host__   close_grid<T>(hipStream_t *streams, robotModel<T> *d_robotModel, gridData<T> *hd_data)",
            "",
            "    __device__ inverse_dynamics_device<T>(T *s_c, const T *s_q, const T *s_qd, const robotModel<T> *d_robotModel, const T gravity)",
            "    __device__ inverse_dynamics_device<T>(T *s_c, const T *s_q, const T *s_qd, const T *s_qdd, const robotModel<T> *d_robotModel, const T gravity)",
            "    __global__ inverse_dynamics_kernel<T>(T *d_c, const T *d_q_qd, const int stride_q_qd, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __global__ inverse_dynamics_kernel<T>(T *d_c, const T *d_q_qd, const int stride_q_qd, const T *d_qdd, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __host__   inverse_dynamics<T,USE_QDD_FLAG=false,USE_COMPRESSED_MEM=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
            "",
            "    __device__ inverse_dynamics_vaf_device<T>(T *s_vaf, const T *s_q, const T *s_qd, const robotModel<T> *d_robotModel, const T gravity)",
            "    __device__ inverse_dynamics_vaf_device<T>(T *s_vaf, const T *s_q, const T *s_qd, const T *s_qdd, const robotModel<T> *d_robotModel, const T gravity)",
            "",
            "    __device__ direct_minv_device<T>(T *s_Minv, const T *s_q, const robotModel<T> *d_robotModel)",
            "    __global__ direct_minv_kernel<T>(T *d_Minv, const T *d_q, const int stride_q, const robotModel<T> *d_robotModel, const int NUM_TIMESTEPS)",
            "    __host__   direct_minv<T,USE_COMPRESSED_MEM=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
            "",
            "    __device__ forward_dynamics_device<T>(T *s_qdd, const T *s_q, const T *s_qd, const T *s_u, const robotModel<T> *d_robotModel, const T gravity)",
            "    __global__ forward_dynamics_kernel<T>(T *d_qdd, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __host__   forward_dynamics<T>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
            "",
            "    __device__ inverse_dynamics_gradient_device<T>(T *s_dc_du, const T *s_q, const T *s_qd, const robotModel<T> *d_robotModel, const T gravity)",
            "    __device__ inverse_dynamics_gradient_device<T>(T *s_dc_du, const T *s_q, const T *s_qd, const T *s_qdd, const robotModel<T> *d_robotModel, const T gravity)",
            "    __global__ inverse_dynamics_gradient_kernel<T>(T *d_dc_du, const T *d_q_qd, const int stride_q_qd, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __global__ inverse_dynamics_gradient_kernel<T>(T *d_dc_du, const T *d_q_qd, const int stride_q_qd, const T *d_qdd, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __host__   inverse_dynamics_gradient<T,USE_QDD_FLAG=false,USE_COMPRESSED_MEM=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
            "",
            "    __device__ forward_dynamics_gradient_device<T>(T *s_df_du, const T *s_q, const T *s_qd, const T *s_u, const robotModel<T> *d_robotModel, const T gravity)",
            "    __device__ forward_dynamics_gradient_device<T>(T *s_df_du, const T *s_q, const T *s_qd, const T *s_qdd, const T *s_Minv, const robotModel<T> *d_robotModel, const T gravity)",
            "    __global__ forward_dynamics_gradient_kernel<T>(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __global__ forward_dynamics_gradient_kernel<T>(T *d_df_du, const T *d_q_qd, const int stride_q_qd, const T *d_qdd, const T *d_Minv, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
            "    __host__   forward_dynamics_gradient<T,USE_QDD_MINV_FLAG=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
            "",
            "    __global__ forward_dynamics_gradient_rollout_kernel<T>(T *d_traj, const T *d_x0, const T *d_u_traj, const T dt, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS, const int NUM_STEPS)",
            "               (no reference counterpart: a consumer of the forward-dynamics gradient -- semi-implicit Euler rollout that writes x+, A, B per step)",
            "",
            "Every host function also exists as <name>_compute_only(...) (device-resident I/O, no streams argument) and",
            "<name>_launch(..., hipStream_t stream) (asynchronous, no synchronisation).",
            "",
            "Suggested Type T is float",
            "",
            "Execution model (differs from the CUDA reference): ONE WAVEFRONT LANE OWNS ONE CONFIGURATION.",
            "    _inner / _device functions take LANE-PRIVATE arrays (registers after inlining), not block-shared memory;",
            "    kernels process 64 consecutive configurations per wavefront and stage the AoS buffers through LDS so",
            "    global loads/stores are coalesced; there is no __syncthreads on the compute path.",
            "Launch kernels with <<<blocks, SUGGESTED_THREADS, <FUNC_CODE>_DYNAMIC_SHARED_MEM_COUNT*sizeof(T)>>> where <FUNC_CODE> = [ID, MINV, FD, ID_DU, FD_DU]",
        ]
        self.gen_add_func_doc("This instance of %s is optimized for the urdf: %s (%d joints), target gfx950 / wave64"
                              % (self.output_file_name(), self.spec.name, n), file_notes)
        self.gen_add_includes(use_thread_group)
        self.gen_add_gpu_err()
        self.gen_add_func_doc("All functions are kept in this namespace")
        self.gen_add_code_line("namespace " + self.file_namespace + " {", True)
        self.gen_add_constants_helpers()
        self.gen_launch_helpers()
        self.gen_lane_helpers()
        self.gen_init_topology_helpers()
        self.gen_init_XImats(include_base_inertia)
        self.gen_init_robotModel()
        self.gen_init_gridData()
        self.gen_load_update_XImats_helpers(use_thread_group)
        self.gen_spatial_algebra_helpers()       # (after the first algorithm-section function: per-function object-cache keys, host.py)
        self.gen_launch_shape_fold()
        self.gen_inverse_dynamics(use_thread_group)
        self.gen_direct_minv(use_thread_group)
        self.gen_forward_dynamics(use_thread_group)
        self.gen_inverse_dynamics_gradient(use_thread_group)
        self.gen_forward_dynamics_gradient(use_thread_group)
        self.gen_forward_dynamics_gradient_coop(use_thread_group)
        self.gen_forward_dynamics_gradient_lean_decl()   # (its kernel is emitted last; the wrappers below dispatch it through this declaration)
        self.gen_forward_dynamics_gradient_host()        # (after the cooperative kernels, which the wrappers dispatch for large robots)
        self.gen_forward_dynamics_gradient_rollout(use_thread_group)
        self.gen_forward_dynamics_gradient_wave(use_thread_group)       # (last: its kernel instance is appended, earlier kernels keep their object-cache keys)
        self.gen_wave_kernels_other(use_thread_group)
        self.gen_forward_dynamics_gradient_lean(use_thread_group)       # (after everything else: no earlier kernel's object-cache key moves)
        self.gen_inverse_dynamics_gradient_lean(use_thread_group)
        self.gen_forward_dynamics_lean(use_thread_group)
        self.gen_direct_minv_lean(use_thread_group)
        self.gen_init_close_grid()
        self.gen_add_end_control_flow()
        self.gen_kernel_instance_list()
        self._report_generation_notes()
        with open(self.output_file_name(), "w") as fh:
            fh.write(self.code_str)
