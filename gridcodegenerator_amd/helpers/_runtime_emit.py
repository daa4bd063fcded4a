"""Emission of the non-algorithmic parts of grid.hip.h: includes, error check, constants, structs,
model-constant upload, gridData allocation, lane math helpers and the wave-level LDS staging that
turns the AoS boundary layout into lane-private registers and back.

Boundary contract reproduced from the reference (names, argument order, buffer layouts):
GRiDCodeGenerator.py:68-114 (constants + structs), :116-153 (init_gridData), :155-203
(init_grid / close_grid), :205-218 (gpuAssert), helpers/_topology_helpers.py:3-54 (init_XImats),
:217-258 (init_topology_helpers), :365-380 (init_robotModel).  The hardware mapping is new.
"""
import numpy as np

WAVE = 64


def _fmt(x):
    x = float(x)
    if x == int(x) and abs(x) < 1e15:
        return "%d" % int(x)
    return repr(x)


class RuntimeEmitMixin:
    # ------------------------------------------------------------------------------------------
    def gen_add_includes(self, use_thread_group=False):
        self.gen_add_code_lines([
            "#pragma once",
            "#include <assert.h>",
            "#include <math.h>",
            "#include <stdio.h>",
            "#include <stdlib.h>",
            "#include <time.h>",
            "#include <hip/hip_runtime.h>",
            "// scheduling fence used inside the straight-line bodies (device code only)",
            "#if defined(__HIP_DEVICE_COMPILE__)",
            "#define GRID_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)",
            "#define GRID_LAUNDER(x) asm volatile(\"\" : \"+v\"(x))   // same value, new identity: defeats CSE of deliberate recomputation",
            "#define GRID_KEEP(x) asm volatile(\"\" :: \"v\"(x))       // the value must exist at this point of the instruction stream",
            "#else",
            "#define GRID_SCHED_FENCE()",
            "#define GRID_LAUNDER(x) ((void)0)",
            "#define GRID_KEEP(x) ((void)(x))",
            "#endif",
            "// single kernel timing helper code",
            "#define time_delta_us_timespec(start,end) (1e6*static_cast<double>(end.tv_sec - start.tv_sec)+1e-3*static_cast<double>(end.tv_nsec - start.tv_nsec))",
            "",
        ])

    def gen_add_gpu_err(self):
        self.gen_add_func_doc("Check for runtime errors using the HIP API",
                              ["Default behaviour mirrors the reference (print, reset, exit).",
                               "Compile with -DGRID_ERRORS_RETURN to record the first error instead of exiting",
                               "(a ctypes / C-ABI caller must survive): query it with grid_first_error()."])
        self.gen_add_code_lines([
            "#ifndef GRID_HIP_ERRCHK_DEFINED",
            "#define GRID_HIP_ERRCHK_DEFINED",
            "inline hipError_t &grid_first_error(){static thread_local hipError_t err = hipSuccess; return err;}",
            "inline const char *&grid_first_error_where(){static thread_local const char *where = \"\"; return where;}",
            "inline int &grid_first_error_line(){static thread_local int line = 0; return line;}",
            "__host__ inline",
            "void gpuAssert(hipError_t code, const char *file, const int line, bool abort=true){",
            "    if (code != hipSuccess){",
            "#ifdef GRID_ERRORS_RETURN",
            "        if (grid_first_error() == hipSuccess){grid_first_error() = code; grid_first_error_where() = file; grid_first_error_line() = line;}",
            "        (void)abort;",
            "#else",
            "        fprintf(stderr,\"GPUassert: %s %s %d\\n\", hipGetErrorString(code), file, line);",
            "        if (abort){hipDeviceReset(); exit(code);}",
            "#endif",
            "    }",
            "}",
            "#define gpuErrchk(err) {gpuAssert(err, __FILE__, __LINE__);}",
            "#endif",
            "",
        ])
        if self.gen_print_mat:
            for const in ("", "const "):
                self.gen_add_code_lines([
                    "template <typename T, int M, int N>",
                    "__host__ __device__",
                    "void printMat(%sT *A, int lda){" % const,
                    "    for(int i=0; i<M; i++){",
                    "        for(int j=0; j<N; j++){printf(\"%.4f \",(double)A[i + lda*j]);}",
                    "        printf(\"\\n\");",
                    "    }",
                    "}",
                    "",
                ])

    # ------------------------------------------------------------------------------------------
    def lds_per_wave(self, alg):
        """LDS elements one wavefront needs for algorithm ``alg`` (input staging vs output chunk)."""
        lay = self.io_layout[alg]
        need = [WAVE * nin for (_, nin) in lay["inputs"]]
        need.append(WAVE * lay["chunk"])
        return max(need) + WAVE * lay.get("table", 0)       # staging region, then the lane-private table (if any)

    @staticmethod
    def _pad(x):
        """Odd per-lane stride => lane L at L*stride hits 64 distinct banks (conflict-free ds_read/ds_write b32)."""
        return x if x % 2 == 1 else x + 1

    def gen_add_constants_helpers(self):
        n = self.spec.n
        waves = self.max_threads // WAVE  # any legal block shape may be launched with the documented LDS size
        self.gen_add_code_lines([
            "const int NUM_JOINTS = %d;" % n,
            "// arithmetic of the kernels for T = float: f32 | f32+f64(Minv,qdd) [mixed] | f64",
            "#define GRID_DTYPE_%s \"%s\"" % (self.file_namespace, {"fp32": "f32", "mixed": "f32+f64(Minv,qdd)", "fp64": "f64"}[self.precision]),
            "// lane-per-configuration: one wavefront (64 lanes) stages 64 configurations through LDS;",
            "// <ALG>_DYNAMIC_SHARED_MEM_COUNT is what a block of up to GRID_MAX_THREADS threads needs (in T elements);",
            "// the host wrappers request only ceil(threads/64) wave regions (grid_lds_bytes)",
            "const int GRID_WAVE_SIZE = %d;" % WAVE,
            "const int ID_DYNAMIC_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("ID")),
            "const int MINV_DYNAMIC_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("MINV")),
            "const int FD_DYNAMIC_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("FD")),
            "const int ID_DU_DYNAMIC_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("ID_DU")),
            "const int FD_DU_DYNAMIC_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("FD_DU")),
            "const int ID_DU_MAX_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("ID_DU")),
            "const int FD_DU_MAX_SHARED_MEM_COUNT = %d;" % (waves * self.lds_per_wave("FD_DU")),
            "const int SUGGESTED_THREADS = %d;" % self.suggested_threads,
            "const int SUGGESTED_MAX_BLOCKS = %d; // 256 CUs x 8; larger batches grid-stride" % self.suggested_max_blocks,
            "const int XIMATS_LANE_COUNT = %d; // per-lane s_XImats: [sin(q_j) | cos(q_j)]" % (2 * n),
            "const int XIMATS_MODEL_COUNT = %d; // d_robotModel->d_XImats: X constants then I, as in the reference" % (72 * n),
            "const int TOPOLOGY_HELPERS_COUNT = %d;" % self.spec.topology_helpers_size(),
        ])
        self.gen_add_code_line("// Define custom structs")
        self.gen_add_code_lines([
            "template <typename T>",
            "struct robotModel {",
            "    T *d_XImats;",
            "    int *d_topology_helpers;",
            "};",
            "template <typename T>",
            "struct gridData {",
            "    // GPU INPUTS",
            "    T *d_q_qd_u;",
            "    T *d_q_qd;",
            "    T *d_q;",
            "    // CPU INPUTS",
            "    T *h_q_qd_u;",
            "    T *h_q_qd;",
            "    T *h_q;",
            "    // GPU OUTPUTS",
            "    T *d_c;",
            "    T *d_Minv;",
            "    T *d_qdd;",
            "    T *d_dc_du;",
            "    T *d_df_du;",
            "    // CPU OUTPUTS",
            "    T *h_c;",
            "    T *h_Minv;",
            "    T *h_qdd;",
            "    T *h_dc_du;",
            "    T *h_df_du;",
            "};",
            "",
        ])

    # ------------------------------------------------------------------------------------------
    def gen_lane_helpers(self):
        """Lane math + accessors + wave-level LDS staging (new in the MI355X design)."""
        compute_f = "double" if self.precision == "fp64" else "float"      # "mixed": C = float, selected regions in D = double
        self.gen_add_code_lines([
            "// ---- lane math: compute type C (generation-time default for T=float: %s) ----" % compute_f,
            "template <typename T> struct grid_compute {typedef T type;};",
            "template <> struct grid_compute<float> {typedef %s type;};" % compute_f,
            "__host__ __device__ __forceinline__ float grid_fma(float a, float b, float c){return __builtin_fmaf(a,b,c);}",
            "__host__ __device__ __forceinline__ double grid_fma(double a, double b, double c){return __builtin_fma(a,b,c);}",
            "template <typename V2> __host__ __device__ __forceinline__ V2 grid_pk_fma(V2 a, V2 b, V2 c){return __builtin_elementwise_fma(a, b, c);}",
            "__host__ __device__ __forceinline__ float grid_rcp(float a){return 1.0f/a;}",
            "__host__ __device__ __forceinline__ double grid_rcp(double a){return 1.0/a;}",
        ])
        if self.trig == "f64":
            self.gen_add_code_lines([
                "// sin/cos evaluated in double then rounded, as the reference does (helpers/_topology_helpers.py:127-128)",
                "__host__ __device__ __forceinline__ void grid_sincos(float x, float *s, float *c){double sd, cd; sincos((double)x, &sd, &cd); *s = (float)sd; *c = (float)cd;}",
            ])
        elif self.trig == "libm":
            self.gen_add_code_lines([
                "__host__ __device__ __forceinline__ void grid_sincos(float x, float *s, float *c){sincosf(x, s, c);}",
            ])
        else:
            self.gen_add_code_lines([
                "// ~25 instructions instead of ~100 for the library sincosf: 3-term Cody-Waite reduction by pi/2 (exact with fma)",
                "// + Cephes minimax polynomials on [-pi/4, pi/4]; max abs error 9.3e-8 for |x| <= 1e5 and 1.0e-7 up to 1e6 (library:",
                "// 7e-8).  BRANCH-FREE on purpose: no lane-divergent control flow anywhere in a generated kernel (DESIGN.md section 9:",
                "// hipcc placed register spills inside the reduced-exec region of exactly such a branch).  The price is the range:",
                "// beyond 1e6 rad (never a joint angle) the reduction loses accuracy (1e-4 at 1e7) and beyond 2.6e7 it is meaningless;",
                "// non-finite input gives NaN like the library.  trig=\"libm\" / \"f64\" keep the library's full-range reduction.",
                "__host__ __device__ __forceinline__ void grid_sincos(float x, float *s, float *c){",
                "    const float k = __builtin_rintf(x*0.636619772367581343f);",
                "    float r = __builtin_fmaf(-k, 1.5707963705062866f, x); r = __builtin_fmaf(-k, -4.371138828673793e-08f, r); r = __builtin_fmaf(-k, -1.7763568394002505e-15f, r);",
                "    const float z = r*r;",
                "    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f); sp = __builtin_fmaf(sp, z, -1.6666654611e-1f); sp = __builtin_fmaf(sp*z, r, r);",
                "    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f); cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);",
                "    cp = __builtin_fmaf(cp*z, z, __builtin_fmaf(z, -0.5f, 1.0f));",
                "    const int q = (int)k;",
                "    const float ss = (q & 1) ? cp : sp; const float cc = (q & 1) ? sp : cp;",
                "    *s = (q & 2) ? -ss : ss; *c = ((q + 1) & 2) ? -cc : cc;",
                "}",
            ])
        self.gen_add_code_lines([
            "__host__ __device__ __forceinline__ void grid_sincos(double x, double *s, double *c){sincos(x, s, c);}",
            "__host__ __device__ __forceinline__ float grid_sin(float x){float s, c; grid_sincos(x, &s, &c); return s;}",
            "__host__ __device__ __forceinline__ float grid_cos(float x){float s, c; grid_sincos(x, &s, &c); return c;}",
            "__host__ __device__ __forceinline__ double grid_sin(double x){return sin(x);}",
            "__host__ __device__ __forceinline__ double grid_cos(double x){return cos(x);}",
            "",
            "/** Block barrier of the tile-cooperative kernels: this wave's LDS writes are complete (lgkmcnt(0)) before it arrives, other",
            " *  waves' LDS reads are issued after they leave.  Like grid_wave_sync it deliberately does NOT wait for outstanding global",
            " *  stores (a workgroup-scope release fence would drain them at every barrier). */",
            "__device__ __forceinline__ void grid_block_sync(){",
            "    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\");",
            "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");",
            "    __builtin_amdgcn_s_barrier();",
            "    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");",
            "}",
            "// ---- accessors the traced cores are written against ----",
            "template <typename T>",
            "struct grid_in_ptrs {",
            "    const T *q_; const T *qd_; const T *u_; const T *qdd_; const T *Minv_; T *tab_; T dt_;",
            "    __host__ __device__ __forceinline__ T dt() const {return dt_;}       // time step of the rollout step cores",
            "    __host__ __device__ __forceinline__ T q(int i) const {return q_[i];}",
            "    __host__ __device__ __forceinline__ T qd(int i) const {return qd_[i];}",
            "    __host__ __device__ __forceinline__ T u(int i) const {return u_[i];}",
            "    __host__ __device__ __forceinline__ T qdd(int i) const {return qdd_[i];}",
            "    __host__ __device__ __forceinline__ T Minv(int i) const {return Minv_[i];}",
            "    // lane-private table of the recomputing gradient cores (a plain local array behind this accessor)",
            "    __host__ __device__ __forceinline__ void tab_put(int i, T v) const {tab_[i] = v;}",
            "    __host__ __device__ __forceinline__ T tab_get(int i) const {return tab_[i];}",
            "};",
            "// the same inputs plus a per-wave LDS table: entry i of lane l lives at i*64 + l (conflict-free).  The kernels of large",
            "// robots park sin q, cos q, qd, qdd there after the prologue and re-load them per gradient column, so that ~4n",
            "// registers are not held for the whole kernel.  Every read goes through a laundered copy of the lane index: hipcc would",
            "// otherwise merge the re-loads of one entry into a single long-lived value (and spill it); a volatile read would do",
            "// too but is compiled to flat_load (address-space inference skips volatile accesses).",
            "template <typename T>",
            "struct grid_in_lds {",
            "    const T *q_; const T *qd_; const T *u_; const T *qdd_; const T *Minv_; T *tab_wave_; int lane_;",
            "    __device__ __forceinline__ T q(int i) const {return q_[i];}",
            "    __device__ __forceinline__ T qd(int i) const {return qd_[i];}",
            "    __device__ __forceinline__ T u(int i) const {return u_[i];}",
            "    __device__ __forceinline__ T qdd(int i) const {return qdd_[i];}",
            "    __device__ __forceinline__ T Minv(int i) const {return Minv_[i];}",
            "    __device__ __forceinline__ void tab_put(int i, T v) const {tab_wave_[i*GRID_WAVE_SIZE + lane_] = v;}",
            "    __device__ __forceinline__ T tab_get(int i) const {int l = lane_; asm volatile(\"\" : \"+v\"(l)); return tab_wave_[i*GRID_WAVE_SIZE + l];}",
            "};",
            "// accessor of the tile-cooperative kernels: the waves of a block work on the SAME 64 configurations and exchange",
            "// per-configuration values through one LDS region (value `slot` of lane l at slot*64 + l; conflict-free).  Reads go through",
            "// a laundered lane index so that every xch_get is its own short-lived load (see grid_in_lds::tab_get).",
            "template <typename T>",
            "struct grid_in_coop {",
            "    const T *q_; const T *qd_; const T *u_; T *xch_; int lane_;",
            "    __device__ __forceinline__ T q(int i) const {return q_[i];}",
            "    __device__ __forceinline__ T qd(int i) const {return qd_[i];}",
            "    __device__ __forceinline__ T u(int i) const {return u_[i];}",
            "    __device__ __forceinline__ void xch_put(int i, T v) const {xch_[i*GRID_WAVE_SIZE + lane_] = v;}",
            "    __device__ __forceinline__ T xch_get(int i) const {int l = lane_; asm volatile(\"\" : \"+v\"(l)); return xch_[i*GRID_WAVE_SIZE + l];}",
            "    __device__ __forceinline__ void barrier() const {grid_block_sync();}",
            "};",
            "// accessors of the two-pass (pipeline) kernels: the per-tile workspace is SoA, value `slot` of lane l at slot*64 + l",
            "template <typename T>",
            "struct grid_in_ws {",
            "    const T *q_; const T *qd_; const T *ws_tile_; mutable int lane_;",
            "    // sync(): launder the lane index so that workspace loads issued after this point cannot be hoisted above it",
            "    // (without it hipcc pulls hundreds of loads to the top of the kernel and spills: 512 registers + 2.7 KB scratch)",
            "    __device__ __forceinline__ void sync() const {asm volatile(\"\" : \"+v\"(lane_));}",
            "    __device__ __forceinline__ T q(int i) const {return q_[i];}",
            "    __device__ __forceinline__ T qd(int i) const {return qd_[i];}",
            "    __device__ __forceinline__ T ws(int slot) const {return ws_tile_[slot*GRID_WAVE_SIZE + lane_];}",
            "};",
            "template <typename T>",
            "struct grid_out_ws {",
            "    T *ws_tile_; int lane_;",
            "    __device__ __forceinline__ void put(int slot, T v){ws_tile_[slot*GRID_WAVE_SIZE + lane_] = v;}",
            "};",
            "template <typename T>",
            "struct grid_out_ptr {",
            "    T *p_;",
            "    __host__ __device__ __forceinline__ void put(int i, T v){p_[i] = v;}",
            "};",
            "// sink of the *_kernel_single_timing latency twins: one lane of the grid writes the (single) result row",
            "template <typename T>",
            "struct grid_out_first {",
            "    T *p_; bool on_;",
            "    __host__ __device__ __forceinline__ void put(int i, T v){if (on_){p_[i] = v;}}    // on_ is WAVE-UNIFORM (every lane of the first wave holds the same row)",
            "};",
            "",
            "// ---- wave-level staging: W lanes <-> W consecutive configurations (W = 64 for whole waves), no block barrier ----",
            "__device__ __forceinline__ void grid_wave_sync(){",
            "    // Orders this wave's LDS writes before its LDS reads across lanes.  DS operations of one wave execute in",
            "    // order, so hardware needs nothing beyond lgkmcnt(0); the wavefront-scope fences + wave_barrier keep the",
            "    // compiler from moving LDS accesses across.  Deliberately NOT a workgroup-scope fence: that adds",
            "    // s_waitcnt vmcnt(0), i.e. a full drain of the previous chunk's global stores (~1-2 us) at every chunk",
            "    // (measured: SQ_WAIT_ANY was 40-50 % of the wave's life).  All LDS traffic here is DS (no FLAT) by construction.",
            "    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\");",
            "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");",
            "    __builtin_amdgcn_wave_barrier();",
            "    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");",
            "}",
            "/** Wave/tile bookkeeping shared by every kernel (flat thread ids; any grid/block shape up to GRID_MAX_THREADS). */",
            "struct grid_tile_iter {",
            "    int lane; int W; int k0_first; int k0_step; int wave_in_block; int part;",
            "    // parts > 1 (column-split kernels): blockIdx % parts selects the column group, blockIdx / parts the tiles;",
            "    // blocks beyond the last full group of `parts` get an empty tile range",
            "    __device__ __forceinline__ grid_tile_iter(const int NUM_TIMESTEPS, const int parts = 1){",
            "        const int nthreads = blockDim.x*blockDim.y*blockDim.z;",
            "        const int tid = threadIdx.x + blockDim.x*(threadIdx.y + blockDim.y*threadIdx.z);",
            "        int nblocks = gridDim.x*gridDim.y*gridDim.z;",
            "        int bid = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z);",
            "        part = bid % parts; bid = bid / parts; nblocks = nblocks / parts;",
            "        lane = tid & (GRID_WAVE_SIZE - 1);",
            "        // wave-uniform by construction; readfirstlane tells the compiler so (SGPRs, scalar loop control)",
            "        wave_in_block = __builtin_amdgcn_readfirstlane(tid / GRID_WAVE_SIZE);",
            "        W = min(GRID_WAVE_SIZE, nthreads - wave_in_block*GRID_WAVE_SIZE); // lanes of this wave (partial last wave allowed)",
            "        k0_first = (bid < nblocks) ? bid*nthreads + wave_in_block*GRID_WAVE_SIZE : NUM_TIMESTEPS;",
            "        k0_step = max(nblocks, 1)*nthreads;",
            "    }",
            "};",
            "/** Opaque copy of a lane index: address arithmetic derived from it is not loop invariant, so LICM cannot hoist the",
            " *  per-element (cfg, i) pairs of the unrolled staging loops out of the tile loop (that pinned ~N registers across",
            " *  the whole straight-line core: Atlas RNEA went from 244 registers to 512 + 222 spills). */",
            "__device__ __forceinline__ int grid_opaque(int x){asm volatile(\"\" : \"+v\"(x)); return x;}",
            "/** The same for a wave-uniform pointer (stays in an SGPR pair): `p + constant` can then be neither hoisted out of the tile loop",
            " *  nor turned into an induction variable -- it is two scalar adds where it is used.",
            " *  The result is a GLOBAL (address space 1) pointer: the round trip through integers would otherwise leave a generic pointer",
            " *  and every access through it would become a flat_ instruction (counted in lgkmcnt as well: the wave-local LDS syncs",
            " *  would then wait for the kernel's global stores). */",
            "#if defined(__HIP_DEVICE_COMPILE__)",
            "#define GRID_GLOBAL __attribute__((address_space(1)))",
            "#else",
            "#define GRID_GLOBAL",
            "#endif",
            "template <typename P> __device__ __forceinline__ GRID_GLOBAL P *grid_opaque_uniform(P *p){",
            "    const unsigned long long a = reinterpret_cast<unsigned long long>(p);",
            "    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));   // (free when already uniform)",
            "    asm volatile(\"\" : \"+s\"(lo), \"+s\"(hi));",
            "    return (GRID_GLOBAL P *)(((unsigned long long)hi << 32) | lo);",
            "}",
            "/** Element at (wave-uniform global pointer) + (32-bit per-lane BYTE offset): the form that maps to `global_* v_off, v_data, s[base]`. */",
            "template <typename P> __device__ __forceinline__ GRID_GLOBAL P &grid_at(GRID_GLOBAL P *base, const unsigned byte_offset){",
            "    return *(GRID_GLOBAL P *)((GRID_GLOBAL char *)base + byte_offset);",
            "}",
            "template <typename P> __device__ __forceinline__ const GRID_GLOBAL P &grid_at(const GRID_GLOBAL P *base, const unsigned byte_offset){",
            "    return *(const GRID_GLOBAL P *)((const GRID_GLOBAL char *)base + byte_offset);",
            "}",
            "__device__ __forceinline__ int grid_opaque_uniform(int x){x = __builtin_amdgcn_readfirstlane(x); asm volatile(\"\" : \"+s\"(x)); return x;}",
            "/**",
            " * Load N values for each of the wave's W configurations (k0 .. k0+W-1, row stride `stride`):",
            " * all N flat coalesced global reads are issued back to back (one memory round trip, not N/4), land in LDS at the",
            " * flat index f = cfg*N + i, and each lane then reads its own row lane*N + i (N even => 2-way bank conflict, cheap).",
            " * Lanes past NUM_TIMESTEPS receive zeros and never store.",
            " */",
            "template <typename T, int N>",
            "__device__ __forceinline__ void grid_load_tile(T *dst, const T *d_src, const int stride, const int k0, const grid_tile_iter &it,",
            "                                               const int NUM_TIMESTEPS, T *s_wave){",
            "    const int nvalid = min(it.W, NUM_TIMESTEPS - k0);",
            "    const GRID_GLOBAL T *src = grid_opaque_uniform(d_src + (size_t)k0*stride);     // wave-uniform base; per-lane offsets stay 32-bit",
            "    const int lane = grid_opaque(it.lane);",
            "    T tmp[N];",
            "    if (it.W == GRID_WAVE_SIZE && nvalid == GRID_WAVE_SIZE && stride == N){",
            "        // whole wave, full tile, dense rows (the common case): the wave's 64*N values are one contiguous block and every",
            "        // index is lane + a compile-time constant (immediate offsets; nothing for the compiler to hoist into SGPRs)",
            "        #pragma unroll",
            "        for (int t = 0; t < N; t++){tmp[t] = grid_at(src + t*GRID_WAVE_SIZE, (unsigned)lane*(unsigned)sizeof(T));}",
            "        #pragma unroll",
            "        for (int t = 0; t < N; t++){s_wave[lane + t*GRID_WAVE_SIZE] = tmp[t];}",
            "    }",
            "    else {",
            "        const int Wo = grid_opaque_uniform(it.W);       // (opaque: the N products t*W are not hoisted into N SGPRs)",
            "        if (nvalid == it.W){    // full tile (wave-uniform): no per-element predicates",
            "            #pragma unroll",
            "            for (int t = 0; t < N; t++){",
            "                const int f = t*Wo + lane; const int cfg = f / N; const int i = f - cfg*N;",
            "                tmp[t] = src[(unsigned)(cfg*stride + i)];",
            "            }",
            "        }",
            "        else {                  // ragged last tile",
            "            #pragma unroll",
            "            for (int t = 0; t < N; t++){",
            "                const int f = t*Wo + lane; const int cfg = f / N; const int i = f - cfg*N;",
            "                const T v = src[(unsigned)(min(cfg, nvalid - 1)*stride + i)];     // clamped read + select: no lane-divergent branch",
            "                tmp[t] = (cfg < nvalid) ? v : static_cast<T>(0);",
            "            }",
            "        }",
            "        #pragma unroll",
            "        for (int t = 0; t < N; t++){s_wave[t*Wo + lane] = tmp[t];}",
            "    }",
            "    grid_wave_sync();",
            "    #pragma unroll",
            "    for (int i = 0; i < N; i++){dst[i] = s_wave[lane*N + i];}",
            "    grid_wave_sync();",
            "}",
            "/**",
            " * Output sink of a kernel: put(i, v) in increasing i.  Values collect in LDS CH at a time and every full",
            " * chunk is written out flat (consecutive lanes -> consecutive addresses inside each configuration's run of",
            " * CH values), so the N_OUT*W results of a wave leave as wide contiguous stores instead of W-way strided ones.",
            " */",
            "// ROW: values per configuration in d_dst.  The NL outputs of this sink are two runs of LEN0 values (NL = 2*LEN0) that",
            "// start at BASE0 and BASE1 of the row (column-split kernels: some d/dq columns and the same d/dqd columns), or one run",
            "// (NL = LEN0, BASE0 = 0) for the whole row.  CH divides LEN0, so a chunk never straddles the two runs.",
            "template <typename T, int ROW, int NL, int CH, int BASE0, int LEN0, int BASE1>",
            "struct grid_out_staged {",
            "    T *s_wave; T *d_dst; int k0; int lane; int W; int NUM_TIMESTEPS;",
            "    // flush: LDS holds the chunk config-major, [cfg][len].  Lanes are split into G = 64/P groups of P = pow2ceil(len)",
            "    // lanes; group g of iteration t writes configuration G*t + g, lane ii of the group element ii.  Every address is",
            "    // (wave-uniform base advanced by a constant per iteration) + (a per-lane offset computed once): no per-element",
            "    // integer arithmetic (the earlier flat f -> (cfg, i) split by division cost ~9 VALU instructions per value:",
            "    // a third of all executed instructions of the Atlas kernels).",
            "    template <int LEN>",
            "    __device__ __forceinline__ void flush_len(const int base){",
            "        constexpr int P = (LEN <= 1) ? 1 : (LEN <= 2) ? 2 : (LEN <= 4) ? 4 : (LEN <= 8) ? 8 : (LEN <= 16) ? 16 : (LEN <= 32) ? 32 : 64;",
            "        constexpr int G = GRID_WAVE_SIZE / P;",
            "        const int nvalid = min(W, NUM_TIMESTEPS - k0);",
            "        // ADDRESSING.  Every address below is (wave-uniform 64-bit base in SGPRs) + (one 32-bit per-lane element offset),",
            "        // and both are derived from OPAQUE copies of k0 and of the lane id taken right here.  Without that, loop strength",
            "        // reduction turns every store address of every flush site (they differ by constants beyond the 4 KB immediate",
            "        // range) into its own 64-bit pointer induction variable of the tile loop: the Atlas-30 gradient kernels carried",
            "        // ~100 such pointer pairs across the whole straight-line core -- 200 registers, most of them living in scratch and",
            "        // read-modify-written at the loop latch (1126 VGPR + 183 SGPR spills, 1.7 KB scratch per lane; DESIGN.md section 9).",
            "        // NO LANE-DIVERGENT CONTROL FLOW.  Every condition below is wave-uniform (scalar branches); lanes that have nothing of",
            "        // their own to write (run length not a power of two, ragged last tile) repeat the last valid element -- same address,",
            "        // same value, merged by the coalescer -- so EXEC never changes inside a kernel.  hipcc placed register spill code inside",
            "        // the reduced-EXEC region of a divergent branch (the spills then did not happen for the masked lanes: DESIGN.md",
            "        // section 9); with no such region there is nowhere to misplace it, and tests/test_kernel_budget.py checks the ISA.",
            "        const int ol = grid_opaque(lane);",
            "        GRID_GLOBAL T *urow = grid_opaque_uniform(d_dst + (size_t)k0*ROW);      // row of the tile's first configuration, wave-uniform",
            "        grid_wave_sync();",
            "        if (W != GRID_WAVE_SIZE){           // partial wavefront (block size not a multiple of 64): generic path, uniform trip count",
            "            const int total = nvalid*LEN;",
            "            for (int f0 = 0; f0 < total; f0 += W){",
            "                const int f = min(f0 + ol, total - 1);",
            "                const int cfg = f / LEN; const int i = f - cfg*LEN;",
            "                urow[(size_t)cfg*ROW + base + i] = s_wave[f];",
            "            }",
            "            grid_wave_sync();",
            "            return;",
            "        }",
            "        if constexpr (LEN % 2 == 0 && ROW % 2 == 0 && sizeof(T) == 4){",
            "            // PAIRS: even run length, even row length, even offset, 8-byte aligned buffer, full tile -> two values per lane and",
            "            // instruction (ds_read_b64 + global_store_dwordx2): half the LDS reads and half the stores.  A lone wavefront pays",
            "            // every LDS round trip and every store issue of a flush in full -- for a 30-joint robot the flushes were a third",
            "            // of the slowest column group's time (profiles/r02/sq_atlas_dID_split4_16384_summary.txt).",
            "            if (nvalid == GRID_WAVE_SIZE && (base & 1) == 0 && (reinterpret_cast<unsigned long long>(d_dst) & 7ull) == 0){",
            "                typedef T T2 __attribute__((ext_vector_type(2)));",
            "                constexpr int L2 = LEN/2;",
            "                constexpr int P2 = (L2 <= 1) ? 1 : (L2 <= 2) ? 2 : (L2 <= 4) ? 4 : (L2 <= 8) ? 8 : (L2 <= 16) ? 16 : (L2 <= 32) ? 32 : 64;",
            "                constexpr int G2 = GRID_WAVE_SIZE / P2;",
            "                constexpr int NIT2 = GRID_WAVE_SIZE / G2;",
            "                const int g2 = ol / P2; const int i2 = min(ol % P2, L2 - 1);",
            "                const T2 *src2 = reinterpret_cast<const T2 *>(s_wave) + (g2*L2 + i2);",
            "                GRID_GLOBAL T2 *ub2 = (GRID_GLOBAL T2 *)(urow + base);",
            "                const unsigned vob2 = (unsigned)(g2*(ROW/2) + i2)*(unsigned)sizeof(T2);",
            "                if constexpr (NIT2 % 8 == 0){",
            "                    for (int t = 0; t < NIT2; t += 8){",
            "                        const T2 *s = src2 + t*G2*L2; GRID_GLOBAL T2 *d = ub2 + (size_t)t*(G2*(ROW/2));",
            "                        const T2 a0 = s[0*G2*L2], a1 = s[1*G2*L2], a2 = s[2*G2*L2], a3 = s[3*G2*L2];",
            "                        const T2 a4 = s[4*G2*L2], a5 = s[5*G2*L2], a6 = s[6*G2*L2], a7 = s[7*G2*L2];",
            "                        grid_at(d + 0*G2*(ROW/2), vob2) = a0; grid_at(d + 1*G2*(ROW/2), vob2) = a1; grid_at(d + 2*G2*(ROW/2), vob2) = a2; grid_at(d + 3*G2*(ROW/2), vob2) = a3;",
            "                        grid_at(d + 4*G2*(ROW/2), vob2) = a4; grid_at(d + 5*G2*(ROW/2), vob2) = a5; grid_at(d + 6*G2*(ROW/2), vob2) = a6; grid_at(d + 7*G2*(ROW/2), vob2) = a7;",
            "                    }",
            "                }",
            "                else {",
            "                    #pragma unroll",
            "                    for (int t = 0; t < NIT2; t++){grid_at(ub2 + (size_t)t*(G2*(ROW/2)), vob2) = src2[t*G2*L2];}",
            "                }",
            "                grid_wave_sync();",
            "                return;",
            "            }",
            "        }",
            "        const int g = ol / P; const int ii = min(ol % P, LEN - 1);",
            "        GRID_GLOBAL T *ubase = urow + base;                  // wave-uniform",
            "        if (nvalid == GRID_WAVE_SIZE){   // full tile",
            "            const T *src = s_wave + (g*LEN + ii);",
            "            const unsigned vob = (unsigned)(g*ROW + ii)*(unsigned)sizeof(T);   // per lane, BYTES, 32 bit: SGPR base + VGPR offset addressing",
            "            constexpr int NIT = GRID_WAVE_SIZE/G;",
            "            if constexpr (NIT % 8 == 0){",
            "                // groups of 8 written out by hand (8 LDS reads in flight, then 8 stores) so that the shape does not",
            "                // depend on the optimisation level (-O1 keeps a `#pragma unroll` loop rolled: one LDS round trip per element)",
            "                for (int t = 0; t < NIT; t += 8){",
            "                    const T *s = src + t*G*LEN; GRID_GLOBAL T *d = ubase + (size_t)t*(G*ROW);     // uniform advance (scalar adds)",
            "                    const T a0 = s[0*G*LEN], a1 = s[1*G*LEN], a2 = s[2*G*LEN], a3 = s[3*G*LEN];",
            "                    const T a4 = s[4*G*LEN], a5 = s[5*G*LEN], a6 = s[6*G*LEN], a7 = s[7*G*LEN];",
            "                    grid_at(d + 0*G*ROW, vob) = a0; grid_at(d + 1*G*ROW, vob) = a1; grid_at(d + 2*G*ROW, vob) = a2; grid_at(d + 3*G*ROW, vob) = a3;",
            "                    grid_at(d + 4*G*ROW, vob) = a4; grid_at(d + 5*G*ROW, vob) = a5; grid_at(d + 6*G*ROW, vob) = a6; grid_at(d + 7*G*ROW, vob) = a7;",
            "                }",
            "            }",
            "            else {",
            "                #pragma unroll",
            "                for (int t = 0; t < NIT; t++){grid_at(ubase + (size_t)t*(G*ROW), vob) = src[t*G*LEN];}",
            "            }",
            "        }",
            "        else {                           // ragged last tile: configurations past the end repeat the last valid one",
            "            for (int t = 0; t*G < nvalid; t++){",
            "                const int cfg = min(t*G + g, nvalid - 1);",
            "                grid_at(ubase, (unsigned)(cfg*ROW + ii)*(unsigned)sizeof(T)) = s_wave[cfg*LEN + ii];",
            "            }",
            "        }",
            "        grid_wave_sync();",
            "    }",
            "    __device__ __forceinline__ void flush(const int chunk){",
            "        const int lbase = chunk*CH;",
            "        const int base = (lbase < LEN0) ? (BASE0 + lbase) : (BASE1 + lbase - LEN0);",
            "        if (NL - lbase < CH){flush_len<(NL % CH == 0) ? CH : (NL % CH)>(base);} else {flush_len<CH>(base);}",
            "    }",
            "    __device__ __forceinline__ void put(const int i, const T v){",
            "        const int lbase = (i / CH)*CH; const int len = (NL - lbase < CH) ? (NL - lbase) : CH;",
            "        s_wave[lane*len + (i % CH)] = v;",
            "        if (((i + 1) % CH) == 0 || i == NL - 1){flush(i / CH);}",
            "    }",
            "};",
            "/**",
            " * Staged sink of a column-split kernel whose block owns an arbitrary SET of gradient columns (COLS...): local values",
            " * arrive column by column (N per column), first the d/dq columns of the set, then the d/dqd columns; chunk k is",
            " * written to N*COLS[k] (k < number of columns) or N*N + N*COLS[k - number of columns].  Same flush as grid_out_staged.",
            " */",
            "template <typename T, int ROW, int N, int... COLS>",
            "struct grid_out_cols {",
            "    T *s_wave; T *d_dst; int k0; int lane; int W; int NUM_TIMESTEPS;",
            "    __device__ __forceinline__ void put(const int i, const T v){",
            "        constexpr int NC = sizeof...(COLS);",
            "        constexpr int cols[NC] = {COLS...};",
            "        s_wave[lane*N + (i % N)] = v;",
            "        if (((i + 1) % N) == 0){",
            "            const int chunk = i / N;",
            "            const int base = (chunk < NC) ? N*cols[chunk] : N*N + N*cols[chunk - NC];",
            "            grid_out_staged<T,ROW,N,N,0,N,0> run = {s_wave, d_dst, k0, lane, W, NUM_TIMESTEPS};",
            "            run.template flush_len<N>(base);",
            "        }",
            "    }",
            "};",
            "/**",
            " * Staged sink whose output row is written in chunks of CH values that may arrive in ANY order: chunk k of the emission",
            " * goes to row offset BASES[k] (rollout step cores: x+, then columns of A and B as the column-serial schedule produces",
            " * them).  The values of chunk 0 are also kept in x_next (the integrator state the lane carries to the next step).",
            " */",
            "template <typename T, int ROW, int CH, int... BASES>",
            "struct grid_out_chunks {",
            "    T *s_wave; T *d_dst; int k0; int lane; int W; int NUM_TIMESTEPS; T *x_next;",
            "    __device__ __forceinline__ void put(const int i, const T v){",
            "        constexpr int NB = sizeof...(BASES);",
            "        constexpr int bases[NB] = {BASES...};",
            "        if (i < CH){x_next[i] = v;}",
            "        s_wave[lane*CH + (i % CH)] = v;",
            "        if (((i + 1) % CH) == 0){",
            "            grid_out_staged<T,ROW,CH,CH,0,CH,0> run = {s_wave, d_dst, k0, lane, W, NUM_TIMESTEPS};",
            "            run.template flush_len<CH>(bases[i / CH]);",
            "        }",
            "    }",
            "};",
            "/**",
            " * Direct output sink: each lane stores its own row (value i at immediate offset 4*i from the lane's row pointer).",
            " * One instruction per value and no LDS round trip; the W-way strided stores are merged by the L2.  Measured on",
            " * MI355X (tools/ubench/staging_floor.hip, 16384 x 98 floats): 5.8 us against 6.2-7.4 us for LDS-staged flat",
            " * copies, whose per-element address arithmetic costs more than it saves.  Used under `if (lane is active)`.",
            " */",
            "template <typename T, int BASE0, int LEN0, int BASE1>",
            "struct grid_out_direct {",
            "    T *row;",
            "    __device__ __forceinline__ void put(const int i, const T v){row[(i < LEN0) ? (BASE0 + i) : (BASE1 + i - LEN0)] = v;}",
            "};",
            "",
        ])

    # ------------------------------------------------------------------------------------------
    def gen_init_topology_helpers(self):
        row = self.spec.topology_helpers_row()
        if not row:
            self.gen_add_code_lines(["//", "// Topology Helpers not needed!", "//",
                                     "template <typename T>", "__host__",
                                     "int *init_topology_helpers(){return nullptr;}", ""])
            return
        self.gen_add_func_doc("Initializes the topology_helpers in GPU memory",
                              ["Same integer table as the reference (parent_inds, [S_inds,] num_ancestors, num_subtree,",
                               "running sums); the lane-per-configuration kernels bake topology into straight-line code and",
                               "never read it -- it is uploaded for API parity and for callers that index with it."],
                              [], "A pointer to the topology_helpers memory in the GPU")
        self.gen_add_code_lines(["template <typename T>", "__host__", "int *init_topology_helpers() {"], True)
        self.gen_add_code_line("int h_topology_helpers[] = {" + ",".join(str(v) for v in row) + "};")
        size = len(row)
        self.gen_add_code_line("int *d_topology_helpers; gpuErrchk(hipMalloc((void**)&d_topology_helpers,%d*sizeof(int)));" % size)
        self.gen_add_code_line("gpuErrchk(hipMemcpy(d_topology_helpers,h_topology_helpers,%d*sizeof(int),hipMemcpyHostToDevice));" % size)
        self.gen_add_code_line("return d_topology_helpers;")
        self.gen_add_end_function()

    def gen_init_XImats(self, include_base_inertia=False):
        table = self.spec.XImats_table()
        self.gen_add_func_doc("Initializes the Xmats and Imats in GPU memory",
                              ["Memory order is X[0...N], I[0...N] (column-major 6x6 blocks; theta-dependent X entries are 0)",
                               "The emitted kernels fold these constants into their instruction streams; the table is uploaded",
                               "for API parity with the reference's robotModel and for user kernels."],
                              [], "A pointer to the XI memory in the GPU")
        self.gen_add_code_lines(["template <typename T>", "__host__", "T* init_XImats() {"], True)
        size = len(table)
        self.gen_add_code_line("static const double h_XImats_d[%d] = {" % size)
        for i in range(0, size, 6):
            self.gen_add_code_line("    " + ", ".join(_fmt(v) for v in table[i:i + 6]) + ("," if i + 6 < size else ""))
        self.gen_add_code_line("};")
        self.gen_add_code_line("T *h_XImats = (T *)malloc(%d*sizeof(T));" % size)
        self.gen_add_code_line("for (int i = 0; i < %d; i++){h_XImats[i] = static_cast<T>(h_XImats_d[i]);}" % size)
        self.gen_add_code_line("T *d_XImats; gpuErrchk(hipMalloc((void**)&d_XImats,%d*sizeof(T)));" % size)
        self.gen_add_code_line("gpuErrchk(hipMemcpy(d_XImats,h_XImats,%d*sizeof(T),hipMemcpyHostToDevice));" % size)
        self.gen_add_code_line("free(h_XImats);")
        self.gen_add_code_line("return d_XImats;")
        self.gen_add_end_function()

    def gen_init_robotModel(self):
        self.gen_add_func_doc("Initializes the robotModel helpers in GPU memory", [], [], "A pointer to the robotModel struct")
        self.gen_add_code_lines(["template <typename T>", "__host__", "robotModel<T>* init_robotModel() {"], True)
        self.gen_add_code_lines([
            "robotModel<T> h_robotModel;",
            "h_robotModel.d_XImats = init_XImats<T>();",
            "h_robotModel.d_topology_helpers = init_topology_helpers<T>();",
            "robotModel<T> *d_robotModel; gpuErrchk(hipMalloc((void**)&d_robotModel,sizeof(robotModel<T>)));",
            "gpuErrchk(hipMemcpy(d_robotModel,&h_robotModel,sizeof(robotModel<T>),hipMemcpyHostToDevice));",
            "return d_robotModel;",
        ])
        self.gen_add_end_function()

    def gen_init_gridData(self):
        body = [
            "gridData<T> *hd_data = (gridData<T> *)malloc(sizeof(gridData<T>));",
            "const size_t K = (size_t)NUM_TIMESTEPS; const size_t N = NUM_JOINTS;",
            "// first the input variables on the GPU",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_q_qd_u, 3*N*K*sizeof(T)));",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_q_qd, 2*N*K*sizeof(T)));",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_q, N*K*sizeof(T)));",
            "// and the CPU (pinned, so the H2D/D2H copies of the host wrappers run at full PCIe rate)",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_q_qd_u, 3*N*K*sizeof(T)));",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_q_qd, 2*N*K*sizeof(T)));",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_q, N*K*sizeof(T)));",
            "// then the GPU outputs",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_c, N*K*sizeof(T)));",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_Minv, N*N*K*sizeof(T)));",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_qdd, N*K*sizeof(T)));",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_dc_du, N*2*N*K*sizeof(T)));",
            "gpuErrchk(hipMalloc((void**)&hd_data->d_df_du, N*2*N*K*sizeof(T)));",
            "// and the CPU",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_c, N*K*sizeof(T)));",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_Minv, N*N*K*sizeof(T)));",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_qdd, N*K*sizeof(T)));",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_dc_du, N*2*N*K*sizeof(T)));",
            "gpuErrchk(hipHostMalloc((void**)&hd_data->h_df_du, N*2*N*K*sizeof(T)));",
            "return hd_data;",
        ]
        self.gen_add_func_doc("Allocated device and host memory for all computations", [], [],
                              "A pointer to the gridData struct of pointers")
        self.gen_add_code_lines(["template <typename T, int NUM_TIMESTEPS>", "__host__", "gridData<T> *init_gridData(){"], True)
        self.gen_add_code_lines(body)
        self.gen_add_end_function()
        self.gen_add_func_doc("Allocated device and host memory for all computations", [],
                              ["Max number of timesteps in the trajectory"], "A pointer to the gridData struct of pointers")
        self.gen_add_code_lines(["template <typename T>", "__host__", "gridData<T> *init_gridData(int NUM_TIMESTEPS){"], True)
        self.gen_add_code_lines(body)
        self.gen_add_end_function()

    def gen_init_close_grid(self):
        MAX_STREAMS = 3
        self.gen_add_func_doc("Initializes streams for host functions", [
            "Nothing to configure for LDS: every kernel needs far less than the 64 KiB default dynamic-LDS limit",
            "(the reference must raise cudaFuncAttributeMaxDynamicSharedMemorySize here, GRiDCodeGenerator.py:164-179)."],
            [], "A pointer to the array of streams")
        self.gen_add_code_lines(["template <typename T>", "__host__", "hipStream_t *init_grid(){"], True)
        self.gen_add_code_lines([
            "hipStream_t *streams = (hipStream_t *)malloc(%d*sizeof(hipStream_t));" % MAX_STREAMS,
            "int priority, minPriority, maxPriority;",
            "gpuErrchk(hipDeviceGetStreamPriorityRange(&minPriority, &maxPriority));",
            "for(int i=0; i<%d; i++){" % MAX_STREAMS,
            "    int adjusted_max = maxPriority - i; priority = adjusted_max > minPriority ? adjusted_max : minPriority;",
            "    gpuErrchk(hipStreamCreateWithPriority(&(streams[i]),hipStreamNonBlocking,priority));",
            "}",
            "return streams;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("Frees the memory used by grid", [
            "Unlike the reference (GRiDCodeGenerator.py:194-202) this also frees d_XImats, d_topology_helpers and hd_data itself."],
            ["streams allocated by init_grid", "robotModel allocated by init_robotModel", "data allocated by init_gridData"], None)
        self.gen_add_code_lines(["template <typename T>", "__host__",
                                 "void close_grid(hipStream_t *streams, robotModel<T> *d_robotModel, gridData<T> *hd_data){"], True)
        self.gen_add_code_lines([
            "if (d_robotModel != nullptr){",
            "    robotModel<T> h_robotModel; gpuErrchk(hipMemcpy(&h_robotModel,d_robotModel,sizeof(robotModel<T>),hipMemcpyDeviceToHost));",
            "    gpuErrchk(hipFree(h_robotModel.d_XImats)); if (h_robotModel.d_topology_helpers != nullptr){gpuErrchk(hipFree(h_robotModel.d_topology_helpers));}",
            "    gpuErrchk(hipFree(d_robotModel));",
            "}",
            "if (hd_data != nullptr){",
            "    gpuErrchk(hipFree(hd_data->d_q_qd_u)); gpuErrchk(hipFree(hd_data->d_q_qd)); gpuErrchk(hipFree(hd_data->d_q));",
            "    gpuErrchk(hipFree(hd_data->d_c)); gpuErrchk(hipFree(hd_data->d_Minv)); gpuErrchk(hipFree(hd_data->d_qdd));",
            "    gpuErrchk(hipFree(hd_data->d_dc_du)); gpuErrchk(hipFree(hd_data->d_df_du));",
            "    gpuErrchk(hipHostFree(hd_data->h_q_qd_u)); gpuErrchk(hipHostFree(hd_data->h_q_qd)); gpuErrchk(hipHostFree(hd_data->h_q));",
            "    gpuErrchk(hipHostFree(hd_data->h_c)); gpuErrchk(hipHostFree(hd_data->h_Minv)); gpuErrchk(hipHostFree(hd_data->h_qdd));",
            "    gpuErrchk(hipHostFree(hd_data->h_dc_du)); gpuErrchk(hipHostFree(hd_data->h_df_du));",
            "    free(hd_data);",
            "}",
            "if (streams != nullptr){for(int i=0; i<%d; i++){gpuErrchk(hipStreamDestroy(streams[i]));} free(streams);}" % MAX_STREAMS,
        ])
        self.gen_add_end_function()
