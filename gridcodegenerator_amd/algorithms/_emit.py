"""Per-algorithm emission: ``gen_<alg>_inner / _device / _kernel / _host / gen_<alg>`` for
inverse dynamics (RNEA), direct Minv, forward dynamics and the two analytical gradients.

Public surface reproduced from the reference (same names / argument order / buffer layouts):
    algorithms/_inverse_dynamics.py:311-495, _direct_minv.py:384-517, _forward_dynamics.py:114-252,
    _inverse_dynamics_gradient.py:652-834, _forward_dynamics_gradient.py:59-242.
What is emitted is different in kind: every ``_inner`` / core is one straight-line body for ONE lane
(trace.py), ``_device`` wraps a fused core over lane-private arrays, ``_kernel`` adds the wave-level
coalesced staging (helpers/_runtime_emit.py) and a grid-stride loop over 64-configuration tiles.
"""
from ..emit import cores, wave
from ..emit.model import SubForest

WAVE = 64
MAX_IN_PIECE = 64   # inputs wider than this are staged through LDS in pieces (keeps LDS/wave small)


def _largest_divisor_leq(n, cap):
    for d in range(min(n, cap), 0, -1):
        if n % d == 0:
            return d
    return 1


class AlgorithmEmitMixin:
    # ------------------------------------------------------------------------------------------
    # layouts
    # ------------------------------------------------------------------------------------------
    def _build_io_layout(self):
        n = self.spec.n
        cap = self.out_chunk

        def chunk(n_out):
            return n_out if n_out <= cap else _largest_divisor_leq(n_out, cap)

        def pieces(total, cap_in=MAX_IN_PIECE):
            return [min(cap_in, total - o) for o in range(0, total, cap_in)]

        lay = {
            "ID": dict(inputs=[("q_qd", 2 * n), ("qdd", n)], n_out=n),
            "MINV": dict(inputs=[("q", n)], n_out=n * n),
            "FD": dict(inputs=[("q_qd_u", 3 * n)], n_out=n),
            "ID_DU": dict(inputs=[("q_qd", 2 * n), ("qdd", n)], n_out=2 * n * n),
            "FD_DU": dict(inputs=[("q_qd_u", 3 * n), ("qdd", n), ("Minv", n * n)], n_out=2 * n * n),
            # rollout consumer: x0 (2n) and u_t (n) in, rows of [x+ | A | B] out in chunks of 2n
            "ROLLOUT": dict(inputs=[("x0", 2 * n), ("u", n)], n_out=cores.rollout_row_count(self.spec)),
        }
        for (alg, d) in lay.items():
            d["chunk"] = chunk(d["n_out"])
            d["in_piece"] = MAX_IN_PIECE
            if alg == "ROLLOUT":
                d["chunk"] = 2 * n      # x+, one column of A or one column of B per flush
            if self.grad_schedule == "recompute" and alg in ("ID_DU", "FD_DU"):
                d["chunk"] = n          # one gradient column per flush
            if self.grad_table and alg in ("ID_DU", "FD_DU"):
                # recomputing column-serial kernels: one gradient column per flush, inputs staged in pieces of <= 32 and a
                # lane-private LDS table behind the staging region (Atlas-30: 64*(32 + 120)*4 B = 38.9 KB per wave, four
                # single-wave blocks per CU)
                d["chunk"] = n
                d["in_piece"] = 32
                d["table"] = cores.recompute_table_size(self.spec, "fd" if alg == "FD_DU" else "id", use_qdd=True)
            d["inputs"] = [(nm, p) for (nm, tot) in d["inputs"] for p in pieces(tot, d["in_piece"])]
        self.io_layout = lay
        # dynamic LDS of one block stays <= 64 KB (no hipFuncSetAttribute needed): cap the block size accordingly
        worst = max(self.lds_per_wave(alg) for alg in lay) * 4
        while self.max_threads > WAVE and (self.max_threads // WAVE) * worst > 65536:
            self.max_threads -= WAVE
        self.suggested_threads = min(self.suggested_threads, self.max_threads)

    # ------------------------------------------------------------------------------------------
    # generic pieces
    # ------------------------------------------------------------------------------------------
    def _emit_traced_function(self, doc, notes, params, template, qualifiers, signature, tracer, store=None, order=None,
                              fence_stmt="GRID_SCHED_FENCE();", read_ahead=0):
        self.gen_add_func_doc(doc, notes, params, None)
        self.gen_add_code_line(template)
        self.gen_add_code_line(qualifiers)
        self.gen_add_code_line(signature + " {", True)
        self.gen_add_code_line("typedef C C2 __attribute__((ext_vector_type(2)));   // packed pair (d/dq, d/dqd): v_pk_* on gfx950")
        self.gen_add_code_line("typedef double D;   // high-precision type of the mixed-precision regions (Minv recursion, qdd = Minv (u - c))")
        ind = "    " * self.indent_level
        lines = tracer.emit(indent=ind, order=order or self.emit_order, store=store, fence_every=self.fence_every, fence_stmt=fence_stmt,
                            read_ahead=read_ahead)
        self.gen_add_raw("\n".join(lines))
        self.gen_add_end_function()
        self.trace_stats[signature.split("(")[0].split()[-1] + "/" + str(len(self.trace_stats))] = tracer.op_counts()

    def _emit_core(self, name, doc, tracer, order=None, fence_stores=True, fence_stmt="GRID_SCHED_FENCE();", read_ahead=0):
        """template <T, C, In, Out> void name(const In &in, Out &out, const T gravity)."""
        self.core_stats[name] = dict(tracer.op_counts(), flops=tracer.flops())
        stride = max(1, int(getattr(self, "fence_stride", 1)))
        counter = [0]

        def fence_after_store():
            # a scheduling fence after every `fence_stride`-th output store (1 = after each: the measured optimum)
            counter[0] += 1
            return " GRID_SCHED_FENCE();" if (fence_stores and counter[0] % stride == 0) else ""
        self._emit_traced_function(
            doc, ["straight-line body for ONE configuration (one wavefront lane); compute type C, storage type T",
                  "in: accessor with q(i), qd(i), u(i), qdd(i), Minv(i); out: sink with put(i, value), i increasing"],
            ["in input accessor", "out output sink", "gravity is the gravity constant"],
            "template <typename T, typename C, typename In, typename Out>",
            "__host__ __device__ __forceinline__",
            "void %s(const In &in, Out &out, const T gravity)" % name, tracer,
            store=lambda dst, val: self._core_store(dst, val, fence_after_store), order=order, fence_stmt=fence_stmt, read_ahead=read_ahead)

    @staticmethod
    def _core_store(dst, val, fence_after_store):
        if isinstance(dst, str):
            if dst.startswith("tab:"):
                return "in.tab_put(%s, (T)(%s));" % (dst[4:], val)
            if dst.startswith("xch:"):
                return "in.xch_put(%s, (T)(%s));" % (dst[4:], val)
            if dst == "barrier":
                return "GRID_SCHED_FENCE(); in.barrier(); GRID_SCHED_FENCE();"
            if dst == "anchor":
                return "GRID_KEEP(%s);" % val
            if dst.startswith("utab:"):
                return "in.utab_put(%s, (T)(%s));" % (dst[5:], val)
            if dst.startswith("utab2:"):
                return "in.utab2_put(%s, (T)(%s));" % (dst[6:], val)
            if dst.startswith("mput:"):
                return "in.m_put(%s, (T)(%s));" % (dst[5:], val)
            if dst == "wsync":
                return "GRID_SCHED_FENCE(); in.sync(); GRID_SCHED_FENCE();"
            if dst.startswith("piece:"):        # (emit/cores.py: AlignedPieces -- value `pos` of a piece of `len` values)
                _, length, pos = dst.split(":")
                return "out.template put_at<%s>(%s, (T)(%s));" % (length, pos, val)
            if dst.startswith("flush:"):
                _, length, base = dst.split(":")
                return "out.template flush<%s>(%s);%s" % (length, base, fence_after_store())
        return "out.put(%s, (T)(%s));%s" % (dst, val, fence_after_store())

    def _emit_load(self, dst, src, total, stride, piece=MAX_IN_PIECE, rows=False):
        if self.in_rows or rows:
            self.gen_add_code_line("grid_rows::load<T,%d>(%s, %s, %s, k0, it, NUM_TIMESTEPS);" % (total, dst, src, stride))
            return
        off = 0
        while off < total:
            p = min(piece, total - off)
            self.gen_add_code_line("grid_load_tile<T,%d>(%s + %d, %s + %d, %s, k0, it, NUM_TIMESTEPS, s_wave);"
                                   % (p, dst, off, src, off, stride))
            off += p

    def _chunk_for(self, length):
        return length if length <= self.out_chunk else _largest_divisor_leq(length, self.out_chunk)

    def _emit_kernel(self, alg, name, core, doc, out_name, primary, extras, has_gravity, accessor, parts=None, chunk=None):
        """primary = (buffer name, count, stride variable); extras = [(buffer name, count)] with row stride = count.
        parts = None: one core writes the whole row.  parts = [(core_name, cols), ...]: column-split kernel, block b
        runs part b % len(parts) on tile group b / len(parts)."""
        n = self.spec.n
        n_out = self.io_layout[alg]["n_out"]
        pname, pcount, pstride = primary
        sig = "void %s(T *d_%s, const T *d_%s, const int %s, " % (name, out_name, pname, pstride)
        for (ename, _) in extras:
            sig += "const T *d_%s, " % ename
        sig += "const robotModel<T> *d_robotModel, " + ("const T gravity, " if has_gravity else "") + "const int NUM_TIMESTEPS)"
        self.kernel_instances.append("__global__ void @NS::%s<T>(%s);" % (
            name, ", ".join(["T *", "const T *", "const int"] + ["const T *"] * len(extras) + ["const @NS::robotModel<T> *"]
                            + (["const T"] if has_gravity else []) + ["const int"])))
        params = ["d_%s is the output buffer, %d values per configuration" % (out_name, n_out),
                  "d_%s is the input buffer, %d values read per configuration" % (pname, pcount),
                  "%s is the stride between configurations in d_%s" % (pstride, pname)]
        params += ["d_%s holds %d values per configuration (dense)" % (en, ec) for (en, ec) in extras]
        params += ["d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)"]
        if has_gravity:
            params.append("gravity is the gravity constant")
        params.append("NUM_TIMESTEPS is the number of configurations")
        notes = ["lane-per-configuration: each wavefront owns 64 consecutive configurations per tile",
                 "launch with <<<blocks, SUGGESTED_THREADS, %s_DYNAMIC_SHARED_MEM_COUNT*sizeof(T)>>>; any block shape up to" % alg,
                 "GRID_MAX_THREADS threads is accepted (the dynamic LDS must cover ceil(threads/64) wave regions)"]
        if parts:
            notes += ["COLUMN-SPLIT variant for small batches: %d column groups %s; the grid's whole wavefronts are numbered block-major and"
                      % (len(parts), [(list(c) if not isinstance(c, tuple) else "dq%s+dqd%s" % (list(c[0]), list(c[1]))) for (_, c) in parts]),
                      "wave gw computes group gw %% %d of tile gw / %d (grid_tile_iter): with %d waves per block a tile's groups share a CU."
                      % (len(parts), len(parts), len(parts)),
                      "Every group repeats the shared prefix (X(q), Minv, RNEA) -- the SIMDs it uses would otherwise idle.",
                      "Blocks of whole wavefronts (use the *_split_launch helper)"]
        self.gen_add_func_doc(doc, notes, params, None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        # the 2-way split is the one used when the batch fills the chip: cap it at 256 registers so two waves share a SIMD
        # (2nd argument = waves per SIMD).  Finer splits only run when there are fewer waves than SIMDs: no cap, no spills.
        # (the finer splits only in the fp32 arithmetic: the double regions of a mixed build would spill 60+ values under the cap)
        occ = 2 if (parts and len(parts) in getattr(self, "split_cap", (2,)) and n <= 12 and (len(parts) == 2 or self.precision == "fp32")) else self.waves_per_simd
        asym = bool(parts) and len(parts) == self.ASYM_SPLIT and all(isinstance(c, tuple) for (_, c) in parts) and n <= 8
        if asym:            # one block of 8 waves per tile: two per SIMD, hence <= 256 registers; per-wave LDS = its largest output half
            self.gen_add_code_line("__global__ __launch_bounds__(%d)" % (WAVE * len(parts)))
        else:
            self.gen_add_code_line("__global__ __launch_bounds__(GRID_MAX_THREADS%s)" % (", %d" % occ if occ > 1 else ""))
        wave_lds = self.split_wave_lds(alg, parts) if asym else self.lds_per_wave(alg)
        self.gen_add_code_line(sig + " {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "const grid_tile_iter it(NUM_TIMESTEPS%s);" % (", %d" % len(parts) if parts else ""),
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % wave_lds,
            "for (int k0 = it.k0_first; k0 < NUM_TIMESTEPS; k0 += it.k0_step){",
        ])
        self.indent_level += 1
        piece = self.io_layout[alg]["in_piece"]
        table = self.io_layout[alg].get("table", 0) if not parts else 0
        self.gen_add_code_line("T s_%s[%d];" % (pname, pcount))
        # the S >= 3 splits of small robots serve batches that leave one wave per SIMD: per-lane 16-byte row loads (grid_rows) save
        # the LDS round trip of the staged path (iiwa-7 K = 16384: dFD 9.97 -> 9.75 us, dID 7.13 -> 6.73 us); kernels that run with
        # many waves per CU keep the coalesced staged loads (K = 1 M: 215 vs 272 us)
        rows = bool(parts) and len(parts) >= 3 and n <= 12
        self._emit_load("s_" + pname, "d_" + pname, pcount, pstride, piece, rows=rows)
        for (ename, ecount) in extras:
            if ecount > MAX_IN_PIECE:
                # large optional inputs (Minv of a 30-joint robot: 900 values) are read where they are used, straight from
                # the lane's row in global memory: staging them would need a 3.6 KB private array per lane
                self.gen_add_code_line("const T *s_%s = d_%s + (size_t)min(k0 + it.lane, NUM_TIMESTEPS - 1)*%d;" % (ename, ename, ecount))
            else:
                self.gen_add_code_line("T s_%s[%d];" % (ename, ecount))
                self._emit_load("s_" + ename, "d_" + ename, ecount, str(ecount), piece)
        if table:
            self.gen_add_code_line("const grid_in_lds<T> in = {%s, s_wave + %d, it.lane};   // table behind the staging region"
                                   % (accessor, self.lds_per_wave(alg) - WAVE * table))
        else:
            self.gen_add_code_line("const grid_in_ptrs<T> in = {%s};" % accessor)
        grav = "gravity" if has_gravity else "static_cast<T>(0)"
        direct = (self.out_mode == "direct")
        if direct:
            self.gen_add_code_line("if (k0 + it.lane < NUM_TIMESTEPS){   // staging above needed every lane; the core does not", True)
            self.gen_add_code_line("T *d_row = d_%s + (size_t)(k0 + it.lane)*%d;" % (out_name, n_out))
        if not parts:
            ch = chunk or self.io_layout[alg]["chunk"]
            assert 64 * ch <= self.lds_per_wave(alg)
            if direct:
                self.gen_add_code_line("grid_out_direct<T,0,%d,0> out = {d_row};" % n_out)
            else:
                self.gen_add_code_line("grid_out_staged<T,%d,%d,%d,0,%d,0> out = {s_wave, d_%s, k0, it.lane, it.W, NUM_TIMESTEPS};"
                                       % (n_out, n_out, ch, n_out, out_name))
            self.gen_add_code_line("%s<T,C>(in, out, %s);" % (core, grav))
        else:
            self.gen_add_code_line("switch (it.part){", True)
            for pi, (pcore, cols) in enumerate(parts):
                if isinstance(cols, tuple):         # ([d/dq columns], [d/dqd columns]): one flush per half
                    lo_cols, hi_cols = cols
                    assert not direct and 64 * n * max(len(lo_cols), len(hi_cols)) <= wave_lds
                    self.gen_add_code_line("case %d: {" % pi, True)
                    self.gen_add_code_line("grid_out_colset2<T,%d,%d,%d,%s> out = {s_wave, d_%s, k0, it.lane, it.W, NUM_TIMESTEPS};   // d/dq columns %s, d/dqd columns %s"
                                           % (n_out, n, len(lo_cols), ",".join(str(c) for c in list(lo_cols) + list(hi_cols)), out_name, list(lo_cols), list(hi_cols)))
                    self.gen_add_code_line("%s<T,C>(in, out, %s);" % (pcore, grav))
                    self.gen_add_code_line("break;")
                    self.gen_add_end_control_flow()
                    continue
                len0 = n * len(cols)
                ch = chunk or self._chunk_for(len0)
                if 64 * ch > self.lds_per_wave(alg):          # (a group of 5 columns of a 12-joint robot: 60 values per half, region sized for 48)
                    ch = _largest_divisor_leq(len0, self.lds_per_wave(alg) // 64)
                assert 64 * ch <= self.lds_per_wave(alg)
                self.gen_add_code_line("case %d: {" % pi, True)
                contiguous = list(cols) == list(range(cols[0], cols[-1] + 1))
                if not contiguous:
                    assert not direct and 64 * n <= self.lds_per_wave(alg)
                    # one flush per half when the whole set fits the wave's LDS region, else one per column
                    if 64 * len0 <= self.lds_per_wave(alg):
                        self.gen_add_code_line("grid_out_colset2<T,%d,%d,%d,%s> out = {s_wave, d_%s, k0, it.lane, it.W, NUM_TIMESTEPS};   // columns %s (both halves)"
                                               % (n_out, n, len(cols), ",".join(str(c) for c in list(cols) + list(cols)), out_name, list(cols)))
                    else:
                        self.gen_add_code_line("grid_out_cols<T,%d,%d,%s> out = {s_wave, d_%s, k0, it.lane, it.W, NUM_TIMESTEPS};   // columns %s"
                                               % (n_out, n, ",".join(str(c) for c in cols), out_name, list(cols)))
                elif direct:
                    self.gen_add_code_line("grid_out_direct<T,%d,%d,%d> out = {d_row};" % (n * cols[0], len0, n * n + n * cols[0]))
                else:
                    self.gen_add_code_line("grid_out_staged<T,%d,%d,%d,%d,%d,%d> out = {s_wave, d_%s, k0, it.lane, it.W, NUM_TIMESTEPS};"
                                           % (n_out, 2 * len0, ch, n * cols[0], len0, n * n + n * cols[0], out_name))
                self.gen_add_code_line("%s<T,C>(in, out, %s);" % (pcore, grav))
                self.gen_add_code_line("break;")
                self.gen_add_end_control_flow()
            self.gen_add_code_line("default: break;")
            self.gen_add_end_control_flow()
        if direct:
            self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        if not parts:
            self._emit_kernel_single_timing(alg, name, core, doc, out_name, primary, extras, has_gravity, accessor)

    def split_wave_lds(self, alg, parts):
        """LDS elements per wave of the asymmetric 8-way split kernel: inputs come by per-lane row loads (no staging), so a wave needs
        only its largest output half (grid_out_colset2)."""
        n = self.spec.n
        return WAVE * n * max(max(len(c[0]), len(c[1])) for (_, c) in parts)

    def _emit_kernel_single_timing(self, alg, name, core, doc, out_name, primary, extras, has_gravity, accessor):
        """Latency twin of a kernel (reference: gen_*_kernel(..., single_call_timing=True), e.g.
        _forward_dynamics_gradient.py:129-131,162-176): same argument list, NUM_TIMESTEPS is the number of REPETITIONS;
        every lane evaluates configuration 0 that many times in a row and one lane writes the result row.  The inputs
        are laundered before every repetition so the compiler can neither hoist nor merge the evaluations.  Not part of
        the explicit-instantiation list: instantiated by whoever calls the *_single_timing host wrapper."""
        n_out = self.io_layout[alg]["n_out"]
        pname, pcount, pstride = primary
        sig = "void %s_single_timing(T *d_%s, const T *d_%s, const int %s, " % (name, out_name, pname, pstride)
        for (ename, _) in extras:
            sig += "const T *d_%s, " % ename
        sig += "const robotModel<T> *d_robotModel, " + ("const T gravity, " if has_gravity else "") + "const int NUM_TIMESTEPS)"
        self.gen_add_func_doc(doc + " -- single-configuration latency twin",
                              ["NUM_TIMESTEPS is overloaded as the number of timing repetitions (as in the reference)",
                               "every lane of the launch evaluates configuration 0; the lanes of block 0's first wave (all holding the same",
                               "values) write d_%s[0..%d) -- a wave-uniform condition, so the kernel has no lane-divergent branch" % (out_name, n_out)],
                              [], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(GRID_MAX_THREADS)")
        self.gen_add_code_line(sig + " {", True)
        self.gen_add_code_line("(void)%s; (void)d_robotModel;" % pstride)
        arrays = [(pname, pcount)] + [(en, ec) for (en, ec) in extras if ec <= MAX_IN_PIECE]
        for (an, ac) in arrays:
            self.gen_add_code_line("T s_%s[%d];" % (an, ac))
            for i0 in range(0, ac, 8):
                self.gen_add_code_line(" ".join("s_%s[%d] = d_%s[%d];" % (an, i, an, i) for i in range(i0, min(ac, i0 + 8))))
        for (en, ec) in extras:
            if ec > MAX_IN_PIECE:
                self.gen_add_code_line("const T *s_%s = d_%s;" % (en, en))
        table = self.io_layout[alg].get("table", 0)
        if table:
            self.gen_add_code_line("T s_tab[%d];" % table)
        self.gen_add_code_line("const grid_in_ptrs<T> in = {%s%s};" % (accessor, ", s_tab" if table else ""))
        self.gen_add_code_line("#if defined(__HIP_DEVICE_COMPILE__)")
        self.gen_add_code_line("const bool first_wave = (blockIdx.x == 0) && (__builtin_amdgcn_readfirstlane((int)threadIdx.x) == 0);   // wave-uniform: scalar branch, EXEC untouched")
        self.gen_add_code_line("#else")
        self.gen_add_code_line("const bool first_wave = true;")
        self.gen_add_code_line("#endif")
        self.gen_add_code_line("grid_out_first<T> out = {d_%s, first_wave};" % out_name)
        self.gen_add_code_line("for (int rep = 0; rep < NUM_TIMESTEPS; rep++){", True)
        for (an, ac) in arrays:
            for i0 in range(0, ac, 8):
                self.gen_add_code_line(" ".join("GRID_LAUNDER(s_%s[%d]);" % (an, i) for i in range(i0, min(ac, i0 + 8))))
        self.gen_add_code_line("%s<T,C>(in, out, %s);" % (core, "gravity" if has_gravity else "static_cast<T>(0)"))
        self.gen_add_end_control_flow()
        self.gen_add_end_function()

    def _emit_pipeline_family(self, alg, base, doc, out_name, primary, has_qdd_variant):
        """Two-pass variant of a gradient kernel for robots whose working set exceeds the register file:
        `<base>_prep_kernel` (RNEA [+ Minv, qdd]) writes v, X a_parent, f, sin/cos [, Minv] to a tile-major SoA workspace,
        `<base>_columns_kernel` runs the column-serial gradient (emit/algorithms.py: rnea_grad_columns) re-reading it."""
        n = self.spec.n
        kind = "fd" if alg == "FD_DU" else "id"
        ws = cores.WorkspaceMap(self.spec, with_minv=(kind == "fd"))
        self.gen_add_code_line("const int %s_WORKSPACE_COUNT = %d; // T elements per configuration of the two-pass workspace "
                               "(allocate ceil(K/64)*64 configurations)" % (alg, ws.count))
        n_out = self.io_layout[alg]["n_out"]
        pname, pcount, pstride = primary
        variants = [("", False)] + ([("_qdd", True)] if has_qdd_variant else [])
        for (sfx, use_qdd) in variants:
            self._emit_core("%s_prep_core%s" % (base, sfx), doc + ": pass 1 (workspace producer)",
                            cores.core_gradient_prep(self.spec, ws, kind, use_qdd), fence_stores=False)
        self._emit_core("%s_columns_core" % base, doc + ": pass 2 (column-serial gradient over the workspace)",
                        cores.core_gradient_columns(self.spec, ws, kind == "fd"), order="creation",
                        fence_stmt="GRID_SCHED_FENCE(); in.sync();")
        # ---- kernels
        for (sfx, use_qdd) in variants:
            name = "%s_prep_kernel" % base
            sig = "void %s(T *d_ws, const T *d_%s, const int %s, %sconst robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)" % (
                name, pname, pstride, "const T *d_qdd, " if use_qdd else "")
            self.kernel_instances.append("__global__ void @NS::%s<T>(%s);" % (
                name, ", ".join(["T *", "const T *", "const int"] + (["const T *"] if use_qdd else []) + ["const @NS::robotModel<T> *", "const T", "const int"])))
            self.gen_add_func_doc(doc + " -- pass 1 of the two-pass variant", ["blocks must be whole wavefronts (threads % 64 == 0)"], [], None)
            self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
            self.gen_add_code_line("__global__ __launch_bounds__(GRID_MAX_THREADS)")
            self.gen_add_code_line(sig + " {", True)
            self.gen_add_code_lines([
                "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
                "const grid_tile_iter it(NUM_TIMESTEPS);",
                "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % self.lds_per_wave(alg),
                "for (int k0 = it.k0_first; k0 < NUM_TIMESTEPS; k0 += it.k0_step){",
            ])
            self.indent_level += 1
            self.gen_add_code_line("T s_%s[%d];" % (pname, pcount))
            self._emit_load("s_" + pname, "d_" + pname, pcount, pstride)
            if use_qdd:
                self.gen_add_code_line("T s_qdd[%d];" % n)
                self._emit_load("s_qdd", "d_qdd", n, str(n))
            acc = "s_%s, s_%s + %d, %s, %s, nullptr" % (pname, pname, n, ("s_%s + %d" % (pname, 2 * n)) if kind == "fd" else "nullptr",
                                                        "s_qdd" if use_qdd else "nullptr")
            self.gen_add_code_line("const grid_in_ptrs<T> in = {%s};" % acc)
            self.gen_add_code_line("grid_out_ws<T> out = {d_ws + (size_t)(k0/GRID_WAVE_SIZE)*%d*GRID_WAVE_SIZE, it.lane};" % ws.count)
            self.gen_add_code_line("%s_prep_core%s<T,C>(in, out, gravity);" % (base, sfx))
            self.gen_add_end_control_flow()
            self.gen_add_end_function()
        name = "%s_columns_kernel" % base
        sig = "void %s(T *d_%s, const T *d_%s, const int %s, const T *d_ws, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)" % (
            name, out_name, pname, pstride)
        self.kernel_instances.append("__global__ void @NS::%s<T>(T *, const T *, const int, const T *, const @NS::robotModel<T> *, const T, const int);" % name)
        self.gen_add_func_doc(doc + " -- pass 2 of the two-pass variant", ["blocks must be whole wavefronts (threads % 64 == 0)"], [], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(GRID_MAX_THREADS)")
        self.gen_add_code_line(sig + " {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "const grid_tile_iter it(NUM_TIMESTEPS);",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % self.lds_per_wave(alg),
            "for (int k0 = it.k0_first; k0 < NUM_TIMESTEPS; k0 += it.k0_step){",
        ])
        self.indent_level += 1
        self.gen_add_code_line("T s_q_qd[%d];" % (2 * n))
        self._emit_load("s_q_qd", "d_" + pname, 2 * n, pstride)
        self.gen_add_code_line("const grid_in_ws<T> in = {s_q_qd, s_q_qd + %d, d_ws + (size_t)(k0/GRID_WAVE_SIZE)*%d*GRID_WAVE_SIZE, it.lane};" % (n, ws.count))
        ch = n
        assert 64 * ch <= self.lds_per_wave(alg)
        self.gen_add_code_line("grid_out_staged<T,%d,%d,%d,0,%d,0> out = {s_wave, d_%s, k0, it.lane, it.W, NUM_TIMESTEPS};"
                               % (n_out, n_out, ch, n_out, out_name))
        self.gen_add_code_line("%s_columns_core<T,C>(in, out, gravity);" % base)
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        # ---- launcher
        self.gen_add_func_doc("Launch the two-pass variant of %s (asynchronous, on `stream`)" % base,
                              ["d_ws: workspace of %s_WORKSPACE_COUNT * ceil(num_timesteps/64)*64 elements" % alg,
                               "threads is rounded up to whole wavefronts"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("void %s_pipeline_launch(T *d_%s, const T *d_%s, const int %s, const T *d_qdd, T *d_ws, const robotModel<T> *d_robotModel, const T gravity,"
                               % (base, out_name, pname, pstride))
        self.gen_add_code_line("        const int num_timesteps, dim3 blocks, dim3 threads, hipStream_t stream) {", True)
        self.gen_add_code_lines([
            "int nthreads = threads.x*threads.y*threads.z; nthreads = ((nthreads + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE)*GRID_WAVE_SIZE;",
            "if (nthreads > GRID_MAX_THREADS){nthreads = GRID_MAX_THREADS;}",
            "threads = dim3(nthreads,1,1); blocks = dim3(blocks.x*blocks.y*blocks.z,1,1);",
            "const size_t lds_bytes = grid_lds_bytes<T>(threads, %d);" % self.lds_per_wave(alg),
        ])
        if has_qdd_variant:
            self.gen_add_code_line("if (d_qdd != nullptr){%s_prep_kernel<T><<<blocks,threads,lds_bytes,stream>>>(d_ws,d_%s,%s,d_qdd,d_robotModel,gravity,num_timesteps);}" % (base, pname, pstride))
            self.gen_add_code_line("else {%s_prep_kernel<T><<<blocks,threads,lds_bytes,stream>>>(d_ws,d_%s,%s,d_robotModel,gravity,num_timesteps);}" % (base, pname, pstride))
        else:
            self.gen_add_code_line("(void)d_qdd; %s_prep_kernel<T><<<blocks,threads,lds_bytes,stream>>>(d_ws,d_%s,%s,d_robotModel,gravity,num_timesteps);" % (base, pname, pstride))
        self.gen_add_code_line("%s_columns_kernel<T><<<blocks,threads,lds_bytes,stream>>>(d_%s,d_%s,%s,d_ws,d_robotModel,gravity,num_timesteps);" % (base, out_name, pname, pstride))
        self.gen_add_end_function()

    def _emit_no_pipeline(self, alg, base, out_name, primary):
        pname, pcount, pstride = primary
        self.gen_add_code_line("const int %s_WORKSPACE_COUNT = 0; // no two-pass variant generated for this robot" % alg)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("void %s_pipeline_launch(T *, const T *, const int, const T *, T *, const robotModel<T> *, const T, const int, dim3, dim3, hipStream_t) {}" % base)
        self.gen_add_code_line("")

    def _split_builder(self, kind):
        """cols -> traced core of a column group.  Recomputing schedule (large robots): every group recomputes only what its
        own columns need, so splitting costs no repeated prefix for dID; dFD keeps the fused builder (Minv is a shared prefix)."""
        if kind == "id" and self.grad_schedule == "recompute":
            def build(cols):
                return cores.core_gradient_recompute(self.spec, "id", cols=cols)
            build.recompute = True
            return build
        if kind == "fd" and self.grad_schedule == "recompute" and self.spec.n > 12:
            def build_fd(cols):
                return cores.core_gradient_recompute(self.spec, "fd", cols=cols)
            build_fd.recompute = True
            return build_fd
        if kind == "id":
            return lambda cols: cores.core_inverse_dynamics_gradient(self.spec, False, cols)
        return lambda cols: cores.core_forward_dynamics_gradient(self.spec, False, cols)

    def _choose_splits(self, builder, displace_first=False):
        """Column-split variants worth emitting: [(S, parts, worst part's op count)], each step improving >= 3 %."""
        n = self.spec.n
        if self.grad_splits == "auto":
            # large robots with the FUSED schedule: the split kernels need > 256 registers per column group, spill, take
            # 10-25 minutes each to compile and were slower than the unsplit kernel (profiles/r01/sweep_atlas30_split.txt)
            # small robots: 2 (full chip, two waves per SIMD), 3 and 4 (K = 16384 is 256 tiles: 4 x 256 = one wave on every
            # SIMD of an MI355X, measured best: 11.5 us vs 12.4 (S=3) vs 17.7 (S=1)) and one column per block for tiny batches.
            # Large robots with the recomputing schedule: 2 and 4, NOT register-capped (a capped build of them spilled ~1000
            # values and faulted); they serve batches of <= 256 / 512 tiles (Atlas-30 at K = 16384 is one wave per CU).
            cand = sorted(set([2, 3, 4, n])) if n <= 8 else ([2, 4] if (n <= 12 or getattr(builder, "recompute", False)) else [])
            limit = 4
        else:
            # entries are split factors S (the columns are grouped automatically) or explicit partitions [[cols], [cols], ...]
            explicit = {len(e): [list(c) for c in e] for e in self.grad_splits if not isinstance(e, int)}
            cand = sorted(set([int(S) for S in self.grad_splits if isinstance(S, int)] + list(explicit)))
            limit = len(cand)
        base = cores._arith_ops(builder(None))
        if not cand:
            return base, []
        cost = cores.range_cost_function(self.spec, builder, exact=(n <= 8))
        if getattr(builder, "recompute", False):
            # column-serial kernels flush one column half at a time, and a lone wavefront pays each flush's LDS round trips and store
            # issue in full: ~200 instruction slots per half with the pair path (measured: the 16-column group of the Atlas-30 dID
            # split was the slowest by 20 us although the groups were balanced on arithmetic).  Balance on arithmetic + flushes.
            arith_cost = cost
            cost = lambda b, e: arith_cost(b, e) + self.FLUSH_SLOTS_PER_COLUMN * (e - b)
        use_sets = (self.split_sets and n <= 8 and self.out_mode == "staged" and not getattr(builder, "recompute", False))
        full = builder(None) if use_sets else None
        picked = []
        last = base
        for S in cand:
            if S > n:
                continue
            if self.grad_splits != "auto" and S in explicit:
                parts = explicit[S]
                assert sorted(c for part in parts for c in part) == list(range(n)), "an explicit partition must cover every column once"
                est = max(cores._arith_ops(builder(c)) for c in parts)
            elif use_sets and S >= 3:
                # arbitrary column sets (exhaustive, exact: cores.optimal_column_sets) for the splits that serve small batches;
                # the 2-way split serves full-chip batches, where contiguous columns give 3-4x longer store runs.  Better still:
                # sets of HALF columns (the d/dq and d/dqd halves of a column share only the prefix) where that lowers the maximum
                parts, est = cores.optimal_column_sets(self.spec, S, full)
                if self.split_half_columns:
                    hparts, hest = cores.optimal_half_column_sets(self.spec, S, full, per_value=self.split_flush_slots)
                    est_cmp = est + (10.0 if self.split_flush_slots == "flush" else self.split_flush_slots) * n * max(2 * len(c_) for c_ in parts)       # (same measure for whole columns)
                    if hest < 0.98 * est_cmp and all(64 * n * max(len(lo), len(hi)) <= self.lds_per_wave("FD_DU") for (lo, hi) in hparts):
                        parts, est = hparts, hest
            else:
                parts, est = cores.balanced_column_split(self.spec, S, cost)
                if displace_first and n <= 8 and 3 <= S < n and self.out_mode == "staged" and not getattr(builder, "recompute", False):
                    # A base joint about the gravity axis makes column 0 almost free (its d/dq half is structurally zero), and a
                    # contiguous split then wastes a whole wave on it (iiwa-7 dID, 4-way: [[0],[1],[2],[3..6]]).  Alternative: split
                    # columns 1.. contiguously and give column 0 to the group it burdens least (one non-contiguous group, flushed
                    # per column).  Measured, iiwa-7 K = 16384: dID 9.5 -> 8.6 us (profiles/r02/exp_iiwa7_column_sets_4way.txt).  NOT
                    # for dFD: same time there, but the kernel needs 262 instead of 256 registers (one wave per SIMD: two streams no
                    # longer overlap, 2.2 -> 1.6 G evals/s) and writes 12 % more (per-column runs): dID only.
                    rest, _ = cores.balanced_column_split(self.spec, S, cost, first=1)
                    if len(rest) == S and all(rest):
                        exact = lambda cols: cores._arith_ops(builder(list(cols)))
                        trials = [[([0] + g if i == k else list(g)) for i, g in enumerate(rest)] for k in range(S)]
                        alt = min(trials, key=lambda t: max(exact(g) for g in t))
                        alt_est = max(exact(g) for g in alt)
                        if alt_est < 0.97 * max(exact(g) for g in parts):
                            # heaviest group first: block b takes group b % S, and the blocks dispatched first should be the ones
                            # that run longest (measured: the same groups in ascending order cost the 4-way dFD split 2 %)
                            parts, est = sorted(alt, key=exact, reverse=True), alt_est
                if getattr(builder, "recompute", False) and n > 12 and len(parts) == S:
                    parts, est = cores.refine_contiguous_split(parts, builder, per_column=self.FLUSH_SLOTS_PER_COLUMN)      # exact costs
            if len(parts) != S or any(not c for c in parts):
                continue
            if self.grad_splits != "auto" or n <= 8 or getattr(builder, "recompute", False) or est < 0.97 * last:
                picked.append((S, parts, est))
                last = est
        if self.grad_splits == "auto" and len(picked) > limit > 1:      # keep the coarsest, the finest and spread the rest
            idx = sorted(set(round(i * (len(picked) - 1) / (limit - 1)) for i in range(limit)))
            picked = [picked[i] for i in idx]
        if use_sets and self.split_half_columns and self.split_asym and 2 * n >= self.ASYM_SPLIT and self.ASYM_SPLIT not in [S for (S, _, _) in picked]:
            # ASYMMETRIC split over the two waves of every SIMD (blocks of 8 waves = one tile on one CU): four heavy groups for the waves
            # dispatched first, which keep the lone-wave pace, four light ones -- d/dqd half-columns need neither the bias torques nor
            # qdd nor the second RNEA pass -- for the waves behind them, whose cost counts split_asym-fold (they get the issue slots
            # the older wave leaves: profiles/r03/two_waves_per_simd.md, 1.5-1.76x slower).  The symmetric 7-way split loses to the
            # 4-way one at K = 16384 because every one of its waves repeats the whole prefix at the pair pace.
            W8 = self.ASYM_SPLIT
            hparts, hest = cores.optimal_half_column_sets(self.spec, W8, full, restarts=60, per_value=self.split_flush_slots,
                                                          weights=[1.0] * (W8 // 2) + [float(self.split_asym)] * (W8 // 2))
            picked.append((W8, hparts, hest))
        chosen = [(S, parts, max(cores._arith_ops(builder(c)) for c in parts)) for (S, parts, _) in picked]
        return base, chosen

    def _emit_split_family(self, alg, kernel_base, core_base, doc, out_name, primary, has_gravity, accessor, builder, launch_args):
        """Cores + kernels + a launcher for the column-split variants of a gradient kernel."""
        base, chosen = self._choose_splits(builder, displace_first=(alg == "ID_DU"))
        self.split_stats[alg] = dict(base_ops=base, splits={S: dict(parts=[(list(c) if not isinstance(c, tuple) else [list(c[0]), list(c[1])])
                                                                            for c in parts], worst_ops=worst)
                                                             for (S, parts, worst) in chosen})
        for (S, parts, worst) in chosen:
            named = []
            for pi, cols in enumerate(parts):
                cname = "%s_s%dp%d" % (core_base, S, pi)
                rec = getattr(builder, "recompute", False)
                what = ("columns %s of d/dq and of d/dqd" % list(cols)) if not isinstance(cols, tuple) else ("columns %s of d/dq, %s of d/dqd" % (list(cols[0]), list(cols[1])))
                self._emit_core(cname, "%s: column group %d of %d (%s)" % (doc, pi, S, what), builder(cols), order="creation" if rec else None,
                                fence_stores=(self.split_fences or S < 3))
                named.append((cname, cols))
            self._emit_kernel(alg, "%s_split%d" % (kernel_base, S), None, doc + " (column-split x%d)" % S, out_name,
                              primary, [], has_gravity, accessor, parts=named, chunk=self.spec.n if rec else None)
        # launcher
        pname, pcount, pstride = primary
        grav = "const T gravity, " if has_gravity else ""
        self.gen_add_func_doc("Launch a column-split variant of %s (asynchronous, on `stream`)" % kernel_base,
                              ["split must be one of %s_SPLITS; tiles_in_flight x split wavefronts are launched (the kernel strides over the" % alg,
                               "remaining tiles), packed into blocks of `threads` threads rounded to whole wavefronts -- 64*split threads put the",
                               "column groups of a tile on one CU; threads.x == 0: that shape (capped at GRID_MAX_THREADS)",
                               "returns false (and launches nothing) for an unsupported split"], [], None)
        self.gen_add_code_line("const int %s_NUM_SPLITS = %d;" % (alg, len(chosen)))
        self.gen_add_code_line("const int %s_SPLITS[%d] = {%s};" % (alg, max(1, len(chosen)), ",".join(str(S) for (S, _, _) in chosen) or "0"))
        self.gen_add_code_line("const int %s_SPLIT_WORST_OPS[%d] = {%s}; // arithmetic ops of the largest part (unsplit: %d)"
                               % (alg, max(1, len(chosen)), ",".join(str(w) for (_, _, w) in chosen) or "0", base))
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool %s_split_launch(const int split, T *d_%s, const T *d_%s, const int %s, const robotModel<T> *d_robotModel, %sconst int num_timesteps,"
                               % (kernel_base.replace("_kernel", ""), out_name, pname, pstride, grav))
        self.gen_add_code_line("        int tiles_in_flight, const dim3 threads, hipStream_t stream) {", True)
        asym_S = [S for (S, parts, _) in chosen if S == self.ASYM_SPLIT and all(isinstance(c, tuple) for c in parts) and self.spec.n <= 8]
        asym_lds = {S: self.split_wave_lds(alg, [(None, c) for c in parts]) for (S, parts, _) in chosen if S in asym_S}
        if asym_S:
            self.gen_add_code_lines([
                "int nthreads = threads.x*threads.y*threads.z;",
                "const bool asym = %s;      // the asymmetric split: ALWAYS one block of `split` waves per tile (heavy groups = the waves dispatched first)" % " || ".join("split == %d" % S for S in asym_S),
                "if (nthreads <= 0 || asym){nthreads = GRID_WAVE_SIZE*split;}",
                "nthreads = ((nthreads + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE)*GRID_WAVE_SIZE; if (nthreads > GRID_MAX_THREADS && !asym){nthreads = GRID_MAX_THREADS;}",
                "const int waves = nthreads/GRID_WAVE_SIZE;",
                "const dim3 block(nthreads,1,1);",
                "size_t lds_bytes = grid_lds_bytes<T>(block, %d);" % self.lds_per_wave(alg),
                "if (tiles_in_flight < 1){tiles_in_flight = 1;}",
                "const dim3 grid((tiles_in_flight*split + waves - 1)/waves,1,1);     // (surplus waves of the last block idle)",
            ])
        else:       # (the text every library without such a kernel has had since round 3: this block is part of every kernel's cache key)
            self.gen_add_code_lines([
                "int nthreads = threads.x*threads.y*threads.z;",
                "if (nthreads <= 0){nthreads = GRID_WAVE_SIZE*split;}",
                "nthreads = ((nthreads + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE)*GRID_WAVE_SIZE; if (nthreads > GRID_MAX_THREADS){nthreads = GRID_MAX_THREADS;}",
                "const int waves = nthreads/GRID_WAVE_SIZE;",
                "const dim3 block(nthreads,1,1);",
                "const size_t lds_bytes = grid_lds_bytes<T>(block, %d);" % self.lds_per_wave(alg),
                "if (tiles_in_flight < 1){tiles_in_flight = 1;}",
                "const dim3 grid((tiles_in_flight*split + waves - 1)/waves,1,1);     // (surplus waves of the last block idle)",
            ])
        self.gen_add_code_line("switch (split){", True)
        for (S, parts, worst) in chosen:
            if S in asym_S:
                assert 4 * S * asym_lds[S] <= 65536
                self.gen_add_code_line("case %d: lds_bytes = grid_lds_bytes<T>(block, %d); %s_split%d<T><<<grid,block,lds_bytes,stream>>>(d_%s,d_%s,%s,d_robotModel,%snum_timesteps); return true;"
                                       % (S, asym_lds[S], kernel_base, S, out_name, pname, pstride, "gravity," if has_gravity else ""))
                continue
            self.gen_add_code_line("case %d: %s_split%d<T><<<grid,block,lds_bytes,stream>>>(d_%s,d_%s,%s,d_robotModel,%snum_timesteps); return true;"
                                   % (S, kernel_base, S, out_name, pname, pstride, "gravity," if has_gravity else ""))
        self.gen_add_code_line("default: return false;")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        # resources of a split kernel (what a dispatch of it really uses: bench.py reports these)
        self.gen_add_func_doc("hipFuncGetAttributes of a column-split variant of %s" % kernel_base, ["returns false for an unsupported split"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool %s_split_attributes(const int split, hipFuncAttributes *attr) {" % kernel_base.replace("_kernel", ""), True)
        self.gen_add_code_line("switch (split){", True)
        for (S, parts, worst) in chosen:
            self.gen_add_code_line("case %d: gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&%s_split%d<T>))); return true;" % (S, kernel_base, S))
        self.gen_add_code_line("default: return false;")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()

    def _emit_host(self, alg, name, doc, template, has_gravity, body_pre, launches, body_post, timing_label):
        """Host wrappers: mode 0 (copies + launch, reference semantics), _compute_only (mode 2) and an
        asynchronous _launch extension (explicit stream, no synchronisation) used by the C-ABI shim."""
        grav = "const T gravity, " if has_gravity else ""
        for mode in ("full", "single_timing", "compute_only", "launch"):
            suffix = {"full": "", "single_timing": "_single_timing", "compute_only": "_compute_only", "launch": "_launch"}[mode]
            tail = {"full": ", hipStream_t *streams", "single_timing": ", hipStream_t *streams", "compute_only": "", "launch": ", hipStream_t stream"}[mode]
            notes = {"full": ["H2D copy of the inputs, kernel, D2H copy of the result, synchronous (reference mode 0)"],
                     "single_timing": ["reference mode 1: ONE configuration (the first of the host buffers) evaluated num_timesteps times inside",
                                       "the *_kernel_single_timing twin, wall-clock per repetition printed as `Single Call %s`" % timing_label],
                     "compute_only": ["inputs/outputs already on the device (reference mode 2: _compute_only), synchronous"],
                     "launch": ["MI355X extension: asynchronous launch on `stream`, no copies, no synchronisation",
                                "(graph-capturable; used by the C-ABI and for multi-GPU sharding on independent streams)"]}[mode]
            self.gen_add_func_doc(doc, notes, ["hd_data is the packaged input and output pointers",
                                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU",
                                               "num_timesteps is the number of configurations",
                                               "block_dimms / thread_dimms: launch shape; illegal shapes are replaced by the suggested one"], None)
            self.gen_add_code_line(template)
            self.gen_add_code_line("__host__")
            self.gen_add_code_line("void %s%s(gridData<T> *hd_data, const robotModel<T> *d_robotModel, %sconst int num_timesteps," % (name, suffix, grav))
            self.gen_add_code_line("        const dim3 block_dimms, const dim3 thread_dimms%s) {" % tail, True)
            single = (mode == "single_timing")
            one = (lambda line: line.replace("*num_timesteps*sizeof(T)", "*sizeof(T)")) if single else (lambda line: line)
            self.gen_add_code_line("dim3 blocks, threads; grid_launch_dims(block_dimms, thread_dimms, %s, &blocks, &threads); grid_fold_launch_z(&blocks, &threads);"
                                   % ("1" if single else "num_timesteps"))
            self.gen_add_code_line("const size_t lds_bytes = grid_lds_bytes<T>(threads, %d);" % self.lds_per_wave(alg))
            stream = {"full": "streams[0]", "single_timing": "streams[0]", "compute_only": "0", "launch": "stream"}[mode]
            for line in body_pre("full" if single else mode):
                self.gen_add_code_line(one(line))
            if mode in ("full", "single_timing"):
                self.gen_add_code_line("gpuErrchk(hipDeviceSynchronize());")
            self.gen_add_code_line("// then call the kernel")
            if single:
                self.gen_add_code_line("struct timespec start, end; clock_gettime(CLOCK_MONOTONIC,&start);")
            for line in launches:
                if single and "@S" in line:
                    continue             # (the latency twins always time the reference-named kernel)
                line = line.replace("_kernel<T>@L", "_kernel_single_timing<T>@L") if single else line
                self.gen_add_code_line(line.replace("@L", "<<<blocks,threads,lds_bytes,%s>>>" % stream).replace("@S", stream))
            self.gen_add_code_line("gpuErrchk(hipGetLastError());")
            if mode != "launch":
                self.gen_add_code_line("gpuErrchk(hipDeviceSynchronize());")
            if single:
                self.gen_add_code_line("clock_gettime(CLOCK_MONOTONIC,&end);")
            if mode in ("full", "single_timing"):
                for line in body_post:
                    self.gen_add_code_line(one(line))
            if single:
                self.gen_add_code_line("printf(\"Single Call %s %%fus\\n\",time_delta_us_timespec(start,end)/static_cast<double>(num_timesteps));" % timing_label)
            self.gen_add_end_function()

    def gen_launch_helpers(self):
        self.gen_add_code_lines([
            "const int GRID_MAX_THREADS = %d; // __launch_bounds__ of every kernel: <= 1 wave per SIMD keeps the full VGPR file" % self.max_threads,
            "/** Sanitise a launch shape: shapes the kernels cannot run (0 or > GRID_MAX_THREADS threads) become the suggested one. */",
            "__host__ inline",
            "void grid_launch_dims(const dim3 block_dimms, const dim3 thread_dimms, const int num_timesteps, dim3 *blocks, dim3 *threads){",
            "    const unsigned long long nthreads = (unsigned long long)thread_dimms.x*thread_dimms.y*thread_dimms.z;",
            "    const unsigned long long nblocks = (unsigned long long)block_dimms.x*block_dimms.y*block_dimms.z;",
            "    if (nthreads == 0 || nthreads > (unsigned long long)GRID_MAX_THREADS || nblocks == 0){",
            "        int b = (num_timesteps + SUGGESTED_THREADS - 1)/SUGGESTED_THREADS; if (b > SUGGESTED_MAX_BLOCKS){b = SUGGESTED_MAX_BLOCKS;} if (b < 1){b = 1;}",
            "        *blocks = dim3(b,1,1); *threads = dim3(SUGGESTED_THREADS,1,1);",
            "    }",
            "    else {*blocks = block_dimms; *threads = thread_dimms;}",
            "}",
            "/** Dynamic LDS bytes for a block of `threads`: one staging region per (possibly partial) wavefront. */",
            "template <typename T>",
            "__host__ inline",
            "size_t grid_lds_bytes(const dim3 threads, const int elems_per_wave){",
            "    const int nthreads = threads.x*threads.y*threads.z;",
            "    return (size_t)((nthreads + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE)*elems_per_wave*sizeof(T);",
            "}",
            "",
        ])

    def gen_launch_shape_fold(self):
        """`grid_fold_launch_z`: the kernels number their threads and blocks by the x and y extents only (the reference's kernels do:
        helpers/_code_generation_helpers.py:41-55); a z extent would give several hardware waves the same thread id -- the same
        staging region, overwritten while another wave flushes it.  The dim3 host wrappers therefore fold z into y (same number of
        threads / blocks, unique flat ids).  Emitted inside the algorithm section: no kernel's object-cache key depends on it."""
        self.gen_add_func_doc("Fold the z extents of a launch shape into y", ["the kernels index threads as x + y*size_x and blocks as x + y*count_x; a z extent",
                                                                            "would repeat ids (two waves sharing one LDS staging region)"], [], None)
        self.gen_add_code_lines(["__host__ inline",
                                 "void grid_fold_launch_z(dim3 *blocks, dim3 *threads){",
                                 "    if (threads->z > 1){threads->y *= threads->z; threads->z = 1;}",
                                 "    if (blocks->z > 1){blocks->y *= blocks->z; blocks->z = 1;}",
                                 "}", ""])

    # ------------------------------------------------------------------------------------------
    # load_update_XImats_helpers (lane-private sin/cos table)
    # ------------------------------------------------------------------------------------------
    def gen_load_update_XImats_helpers_temp_mem_size(self):
        return 0

    def gen_load_update_XImats_helpers(self, use_thread_group=False):
        n = self.spec.n
        self.gen_add_func_doc("Updates the lane-private X(q) cache according to the configuration",
                              ["The reference copies 72n model constants to shared memory and patches the theta-dependent",
                               "entries (helpers/_topology_helpers.py:90-182).  Here the constants live in the instruction stream,",
                               "so all that depends on q is sin/cos: s_XImats[j] = sin(q_j), s_XImats[NUM_JOINTS + j] = cos(q_j)."],
                              ["s_XImats is the lane-private destination of size XIMATS_LANE_COUNT = " + str(2 * n),
                               "s_q is the lane-private vector of joint positions",
                               "d_robotModel is unused (API parity)", "s_temp is unused (API parity; may be nullptr)"], None)
        self.gen_add_code_lines(["template <typename T>", "__host__ __device__ __forceinline__",
                                 "void load_update_XImats_helpers(T *s_XImats, const T *s_q, const robotModel<T> *d_robotModel, T *s_temp) {"], True)
        self.gen_add_code_line("(void)d_robotModel; (void)s_temp;")
        for j in range(n):
            if self.spec.uses_trig[j]:
                self.gen_add_code_line("{typename grid_compute<T>::type s, c; grid_sincos((typename grid_compute<T>::type)s_q[%d], &s, &c); "
                                       "s_XImats[%d] = (T)s; s_XImats[%d] = (T)c;}" % (j, j, n + j))
            else:
                self.gen_add_code_line("s_XImats[%d] = static_cast<T>(0); s_XImats[%d] = static_cast<T>(1);" % (j, n + j))
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # inverse dynamics
    # ------------------------------------------------------------------------------------------
    def gen_inverse_dynamics_inner_temp_mem_size(self):
        return 0

    def gen_inverse_dynamics_inner(self, use_thread_group=False, compute_c=True, use_qdd_input=True):
        tr = cores.inner_inverse_dynamics(self.spec, compute_c, use_qdd_input)
        name = "inverse_dynamics_inner" if compute_c else "inverse_dynamics_inner_vaf"
        sig = "void %s(%sT *s_vaf, const T *s_q, const T *s_qd, %sT *s_XImats, T *s_temp, const T gravity)" % (
            name, "T *s_c,  " if compute_c else "", "const T *s_qdd, " if use_qdd_input else "")
        self._emit_traced_function("Compute the RNEA (Recursive Newton-Euler Algorithm)",
                                   ([] if use_qdd_input else ["optimized for qdd = 0"]) + ["lane-private pointers; s_temp unused"],
                                   ["s_vaf receives v, a, f (18*NUM_JOINTS)", "s_XImats is the lane-private sin/cos table"],
                                   "template <typename T, typename C = typename grid_compute<T>::type>",
                                   "__host__ __device__ __forceinline__", sig, tr)

    def gen_inverse_dynamics_device(self, use_thread_group=False, compute_c=True, use_qdd_input=True):
        n = self.spec.n
        name = "inverse_dynamics_device" if compute_c else "inverse_dynamics_vaf_device"
        core = ("inverse_dynamics_core" if compute_c else "inverse_dynamics_vaf_core") + ("_qdd" if use_qdd_input else "")
        out = "s_c" if compute_c else "s_vaf"
        sig = "void %s(T *%s, const T *s_q, const T *s_qd, %sconst robotModel<T> *d_robotModel, const T gravity)" % (
            name, out, "const T *s_qdd, " if use_qdd_input else "")
        self.gen_add_func_doc("Compute the RNEA (Recursive Newton-Euler Algorithm)",
                              ["lane-private arrays in, lane-private array out"], [], None)
        self.gen_add_code_lines(["template <typename T, typename C = typename grid_compute<T>::type>",
                                 "__host__ __device__ __forceinline__", sig + " {"], True)
        self.gen_add_code_lines(["(void)d_robotModel;",
                                 "const grid_in_ptrs<T> in = {s_q, s_qd, nullptr, %s, nullptr};" % ("s_qdd" if use_qdd_input else "nullptr"),
                                 "grid_out_ptr<T> out = {%s};" % out,
                                 "%s<T,C>(in, out, gravity);" % core])
        self.gen_add_end_function()

    def gen_inverse_dynamics_kernel(self, use_thread_group=False, use_qdd_input=False, single_call_timing=False):
        n = self.spec.n
        self._emit_kernel("ID", "inverse_dynamics_kernel", "inverse_dynamics_core" + ("_qdd" if use_qdd_input else ""),
                          "Compute the RNEA (Recursive Newton-Euler Algorithm)", "c", ("q_qd", 2 * n, "stride_q_qd"),
                          [("qdd", n)] if use_qdd_input else [], True,
                          "s_q_qd, s_q_qd + %d, nullptr, %s, nullptr" % (n, "s_qdd" if use_qdd_input else "nullptr"))

    def gen_inverse_dynamics_host(self, mode=0):
        def pre(mode):
            if mode == "full":
                return ["int stride_q_qd;",
                        "if (USE_COMPRESSED_MEM) {stride_q_qd = 2*NUM_JOINTS; gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd,hd_data->h_q_qd,stride_q_qd*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                        "else {stride_q_qd = 3*NUM_JOINTS; gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                        "if (USE_QDD_FLAG) {gpuErrchk(hipMemcpyAsync(hd_data->d_qdd,hd_data->h_qdd,NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[1]));}"]
            return ["const int stride_q_qd = USE_COMPRESSED_MEM ? 2*NUM_JOINTS: 3*NUM_JOINTS;"]
        launches = [
            "const T *d_in = USE_COMPRESSED_MEM ? hd_data->d_q_qd : hd_data->d_q_qd_u;",
            "if (USE_QDD_FLAG) {inverse_dynamics_kernel<T>@L(hd_data->d_c,d_in,stride_q_qd,hd_data->d_qdd,d_robotModel,gravity,num_timesteps);}",
            "else              {inverse_dynamics_kernel<T>@L(hd_data->d_c,d_in,stride_q_qd,d_robotModel,gravity,num_timesteps);}"]
        post = ["// finally transfer the result back",
                "gpuErrchk(hipMemcpy(hd_data->h_c,hd_data->d_c,NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyDeviceToHost));",
                "gpuErrchk(hipDeviceSynchronize());"]
        self._emit_host("ID", "inverse_dynamics", "Compute the RNEA (Recursive Newton-Euler Algorithm)",
                        "template <typename T, bool USE_QDD_FLAG = false, bool USE_COMPRESSED_MEM = false>", True, pre, launches, post, "ID")

    def gen_inverse_dynamics(self, use_thread_group=False):
        for use_qdd in (True, False):
            self._emit_core("inverse_dynamics_core" + ("_qdd" if use_qdd else ""),
                            "RNEA core: c = ID(q, qd%s)" % (", qdd" if use_qdd else ", 0"),
                            cores.core_inverse_dynamics(self.spec, use_qdd))
            self._emit_core("inverse_dynamics_vaf_core" + ("_qdd" if use_qdd else ""),
                            "RNEA core returning v, a, f (18*NUM_JOINTS values: v | a | f, 6 per joint)",
                            cores.core_inverse_dynamics_vaf(self.spec, use_qdd))
        if self.emit_inner_api:
            for compute_c in (True, False):
                for use_qdd in (True, False):
                    self.gen_inverse_dynamics_inner(use_thread_group, compute_c, use_qdd)
        for compute_c in (True, False):
            for use_qdd in (True, False):
                self.gen_inverse_dynamics_device(use_thread_group, compute_c, use_qdd)
        self.gen_inverse_dynamics_kernel(use_thread_group, True)
        self.gen_inverse_dynamics_kernel(use_thread_group, False)
        self.gen_inverse_dynamics_host()

    # ------------------------------------------------------------------------------------------
    # direct Minv
    # ------------------------------------------------------------------------------------------
    def gen_direct_minv_inner_temp_mem_size(self):
        return 0

    def gen_direct_minv_inner(self, use_thread_group=False):
        self._emit_traced_function("Compute the inverse of the mass matrix",
                                   ["Outputs a SYMMETRIC_UPPER triangular matrix for Minv (lower half written as 0)",
                                    "lane-private pointers; s_temp unused"],
                                   ["s_Minv is a pointer to memory for the final result (NUM_JOINTS^2, column-major)"],
                                   "template <typename T, typename C = typename grid_compute<T>::type>",
                                   "__host__ __device__ __forceinline__",
                                   "void direct_minv_inner(T *s_Minv, const T *s_q, T *s_XImats, T *s_temp)",
                                   cores.inner_direct_minv(self.spec))

    def gen_direct_minv_device(self, use_thread_group=False):
        self.gen_add_func_doc("Compute the inverse of the mass matrix", ["Outputs a SYMMETRIC_UPPER triangular matrix for Minv"], [], None)
        self.gen_add_code_lines(["template <typename T, typename C = typename grid_compute<T>::type>",
                                 "__host__ __device__ __forceinline__",
                                 "void direct_minv_device(T *s_Minv, const T *s_q, const robotModel<T> *d_robotModel){"], True)
        self.gen_add_code_lines(["(void)d_robotModel;",
                                 "const grid_in_ptrs<T> in = {s_q, nullptr, nullptr, nullptr, nullptr};",
                                 "grid_out_ptr<T> out = {s_Minv};",
                                 "direct_minv_core<T,C>(in, out, static_cast<T>(0));"])
        self.gen_add_end_function()

    def gen_direct_minv_kernel(self, use_thread_group=False, single_call_timing=False):
        n = self.spec.n
        self._emit_kernel("MINV", "direct_minv_kernel", "direct_minv_core", "Compute the inverse of the mass matrix",
                          "Minv", ("q", n, "stride_q"), [], False, "s_q, nullptr, nullptr, nullptr, nullptr")

    def gen_direct_minv_host(self, mode=0):
        def pre(mode):
            if mode == "full":
                return ["int stride_q;",
                        "if (USE_COMPRESSED_MEM) {stride_q = NUM_JOINTS; gpuErrchk(hipMemcpyAsync(hd_data->d_q,hd_data->h_q,stride_q*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                        "else {stride_q = 3*NUM_JOINTS; gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));}"]
            return ["const int stride_q = USE_COMPRESSED_MEM ? NUM_JOINTS: 3*NUM_JOINTS;"]
        launches = [
            "if (false) {}",
            "else if (MINV_LEAN_AUTO_MIN_TILES > 0 && (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE >= MINV_LEAN_AUTO_MIN_TILES && "
            "(MINV_LEAN_AUTO_MAX_TILES == 0 || (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE <= MINV_LEAN_AUTO_MAX_TILES) && "
            "direct_minv_lean_launch<T>(hd_data->d_Minv,USE_COMPRESSED_MEM ? hd_data->d_q : hd_data->d_q_qd_u,stride_q,d_robotModel,num_timesteps,0,@S)) {}",
            "else {direct_minv_kernel<T>@L(hd_data->d_Minv,USE_COMPRESSED_MEM ? hd_data->d_q : hd_data->d_q_qd_u,stride_q,d_robotModel,num_timesteps);}"]
        post = ["// finally transfer the result back",
                "gpuErrchk(hipMemcpy(hd_data->h_Minv,hd_data->d_Minv,NUM_JOINTS*NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyDeviceToHost));",
                "gpuErrchk(hipDeviceSynchronize());"]
        self._emit_host("MINV", "direct_minv", "Compute the inverse of the mass matrix",
                        "template <typename T, bool USE_COMPRESSED_MEM = false>", False, pre, launches, post, "Minv")

    def gen_direct_minv(self, use_thread_group=False):
        self._emit_core("direct_minv_core", "Direct (Carpentier) inverse of the joint-space inertia matrix, upper triangle",
                        cores.core_direct_minv(self.spec))
        if self.emit_inner_api:
            self.gen_direct_minv_inner(use_thread_group)
        self.gen_direct_minv_device(use_thread_group)
        self.gen_direct_minv_kernel(use_thread_group)
        self.gen_direct_minv_lean_decl()
        self.gen_direct_minv_host()

    # ------------------------------------------------------------------------------------------
    # forward dynamics
    # ------------------------------------------------------------------------------------------
    def gen_forward_dynamics_inner_temp_mem_size(self):
        return 0

    def gen_forward_dynamics_finish(self, use_thread_group=False):
        self._emit_traced_function("Finish the forward dynamics computation with qdd = Minv*(u-c)", [],
                                   ["s_qdd is a pointer to memory for the final result", "s_u is the vector of joint input torques",
                                    "s_c is the bias vector", "s_Minv is the (upper triangular, column-major) inverse mass matrix"],
                                   "template <typename T, typename C = typename grid_compute<T>::type>",
                                   "__host__ __device__ __forceinline__",
                                   "void forward_dynamics_finish(T *s_qdd, const T *s_u, const T *s_c, const T *s_Minv)",
                                   cores.inner_forward_dynamics_finish(self.spec))

    def gen_forward_dynamics_inner(self, use_thread_group=False):
        self._emit_traced_function("Computes forward dynamics", ["lane-private pointers; s_temp unused"],
                                   ["s_qdd is a pointer to memory for the final result"],
                                   "template <typename T, typename C = typename grid_compute<T>::type>",
                                   "__host__ __device__ __forceinline__",
                                   "void forward_dynamics_inner(T *s_qdd, const T *s_q, const T *s_qd, const T *s_u, T *s_XImats, T *s_temp, const T gravity)",
                                   cores.inner_forward_dynamics(self.spec))

    def gen_forward_dynamics_device(self, use_thread_group=False):
        self.gen_add_func_doc("Computes forward dynamics", [], [], None)
        self.gen_add_code_lines(["template <typename T, typename C = typename grid_compute<T>::type>",
                                 "__host__ __device__ __forceinline__",
                                 "void forward_dynamics_device(T *s_qdd, const T *s_q, const T *s_qd, const T *s_u, const robotModel<T> *d_robotModel, const T gravity) {"], True)
        self.gen_add_code_lines(["(void)d_robotModel;",
                                 "const grid_in_ptrs<T> in = {s_q, s_qd, s_u, nullptr, nullptr};",
                                 "grid_out_ptr<T> out = {s_qdd};",
                                 "forward_dynamics_core<T,C>(in, out, gravity);"])
        self.gen_add_end_function()

    def gen_forward_dynamics_kernel(self, use_thread_group=False, single_call_timing=False):
        n = self.spec.n
        self._emit_kernel("FD", "forward_dynamics_kernel", "forward_dynamics_core", "Computes forward dynamics",
                          "qdd", ("q_qd_u", 3 * n, "stride_q_qd_u"), [], True,
                          "s_q_qd_u, s_q_qd_u + %d, s_q_qd_u + %d, nullptr, nullptr" % (n, 2 * n))

    def gen_forward_dynamics_host(self, mode=0):
        def pre(mode):
            lines = ["const int stride_q_qd_u = 3*NUM_JOINTS;"]
            if mode == "full":
                lines.append("gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd_u*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));")
            return lines
        launches = [
            # large robots: the register-lean tile-cooperative kernel where the generator emitted one (FD_LEAN_AUTO_MIN_TILES) -- same
            # outputs to round-off, blocks/threads then unused
            "if (false) {}",
            "else if (FD_LEAN_AUTO_MIN_TILES > 0 && (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE >= FD_LEAN_AUTO_MIN_TILES && "
            "(FD_LEAN_AUTO_MAX_TILES == 0 || (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE <= FD_LEAN_AUTO_MAX_TILES) && "
            "forward_dynamics_lean_launch<T>(hd_data->d_qdd,hd_data->d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps,0,@S)) {}",
            "else {forward_dynamics_kernel<T>@L(hd_data->d_qdd,hd_data->d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);}"]
        post = ["// finally transfer the result back",
                "gpuErrchk(hipMemcpy(hd_data->h_qdd,hd_data->d_qdd,NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyDeviceToHost));",
                "gpuErrchk(hipDeviceSynchronize());"]
        self._emit_host("FD", "forward_dynamics", "Computes forward dynamics", "template <typename T>", True, pre, launches, post, "FD")

    def gen_forward_dynamics(self, use_thread_group=False):
        self._emit_core("forward_dynamics_core", "Forward dynamics core: qdd = Minv(q) (u - c(q, qd))",
                        cores.core_forward_dynamics(self.spec))
        if self.emit_inner_api:
            self.gen_forward_dynamics_finish(use_thread_group)
            self.gen_forward_dynamics_inner(use_thread_group)
        self.gen_forward_dynamics_device(use_thread_group)
        self.gen_forward_dynamics_kernel(use_thread_group)
        self.gen_forward_dynamics_lean_decl()
        self.gen_forward_dynamics_host()

    # ------------------------------------------------------------------------------------------
    # inverse dynamics gradient
    # ------------------------------------------------------------------------------------------
    def gen_inverse_dynamics_gradient_inner_temp_mem_size(self):
        return 0

    def gen_inverse_dynamics_gradient_kernel_max_temp_mem_size(self):
        return 0

    def gen_inverse_dynamics_gradient_inner(self, use_thread_group=False):
        self._emit_traced_function("Computes the gradient of inverse dynamics",
                                   ["Uses the precomputed v, a, f in s_vaf (f accumulated over subtrees)", "lane-private pointers; s_temp unused"],
                                   ["s_dc_du is a pointer to memory for the final result of size 2*NUM_JOINTS*NUM_JOINTS"],
                                   "template <typename T, typename C = typename grid_compute<T>::type>",
                                   "__host__ __device__ __forceinline__",
                                   "void inverse_dynamics_gradient_inner(T *s_dc_du, const T *s_q, const T *s_qd, const T *s_vaf, T *s_XImats, T *s_temp, const T gravity)",
                                   cores.inner_inverse_dynamics_gradient_columns(self.spec) if self.grad_schedule == "recompute"
                                   else cores.inner_inverse_dynamics_gradient(self.spec),
                                   order="creation" if self.grad_schedule == "recompute" else None)

    def gen_inverse_dynamics_gradient_device(self, use_thread_group=False, use_qdd_input=False):
        self.gen_add_func_doc("Computes the gradient of inverse dynamics", [] if use_qdd_input else ["optimized for qdd = 0"], [], None)
        self.gen_add_code_lines(["template <typename T, typename C = typename grid_compute<T>::type>",
                                 "__host__ __device__ __forceinline__",
                                 "void inverse_dynamics_gradient_device(T *s_dc_du, const T *s_q, const T *s_qd, %sconst robotModel<T> *d_robotModel, const T gravity) {"
                                 % ("const T *s_qdd, " if use_qdd_input else "")], True)
        tab = self.io_layout["ID_DU"].get("table", 0)
        if tab:
            self.gen_add_code_line("T s_tab[%d];   // lane-private table of the recomputing core" % tab)
        self.gen_add_code_lines(["(void)d_robotModel;",
                                 "const grid_in_ptrs<T> in = {s_q, s_qd, nullptr, %s, nullptr%s};" % ("s_qdd" if use_qdd_input else "nullptr", ", s_tab" if tab else ""),
                                 "grid_out_ptr<T> out = {s_dc_du};",
                                 "inverse_dynamics_gradient_core%s<T,C>(in, out, gravity);" % ("_qdd" if use_qdd_input else "")])
        self.gen_add_end_function()

    def gen_inverse_dynamics_gradient_kernel(self, use_thread_group=False, use_qdd_input=False, single_call_timing=False):
        n = self.spec.n
        self._emit_kernel("ID_DU", "inverse_dynamics_gradient_kernel", "inverse_dynamics_gradient_core" + ("_qdd" if use_qdd_input else ""),
                          "Computes the gradient of inverse dynamics", "dc_du", ("q_qd", 2 * n, "stride_q_qd"),
                          [("qdd", n)] if use_qdd_input else [], True,
                          "s_q_qd, s_q_qd + %d, nullptr, %s, nullptr" % (n, "s_qdd" if use_qdd_input else "nullptr"),
                          chunk=n if self.grad_schedule == "recompute" else None)

    def gen_inverse_dynamics_gradient_host(self, mode=0):
        def pre(mode):
            if mode == "full":
                return ["int stride_q_qd;",
                        "if (USE_COMPRESSED_MEM) {stride_q_qd = 2*NUM_JOINTS; gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd,hd_data->h_q_qd,stride_q_qd*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                        "else {stride_q_qd = 3*NUM_JOINTS; gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));}",
                        "if (USE_QDD_FLAG) {gpuErrchk(hipMemcpyAsync(hd_data->d_qdd,hd_data->h_qdd,NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[1]));}"]
            return ["const int stride_q_qd = USE_COMPRESSED_MEM ? 2*NUM_JOINTS: 3*NUM_JOINTS;"]
        launches = [
            "const T *d_in = USE_COMPRESSED_MEM ? hd_data->d_q_qd : hd_data->d_q_qd_u;",
            "if (USE_QDD_FLAG) {inverse_dynamics_gradient_kernel<T>@L(hd_data->d_dc_du,d_in,stride_q_qd,hd_data->d_qdd,d_robotModel,gravity,num_timesteps);}",
            # large robots at qdd = 0: the register-lean tile-cooperative kernel where the generator emitted one (ID_DU_LEAN_AUTO_MIN_TILES:
            # from how many tiles on) -- same outputs, blocks/threads then unused
            "else if (ID_DU_LEAN_AUTO_MIN_TILES > 0 && (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE >= ID_DU_LEAN_AUTO_MIN_TILES && "
            "inverse_dynamics_gradient_lean_launch<T>(hd_data->d_dc_du,d_in,stride_q_qd,d_robotModel,gravity,num_timesteps,0,@S)) {}",
            "else              {inverse_dynamics_gradient_kernel<T>@L(hd_data->d_dc_du,d_in,stride_q_qd,d_robotModel,gravity,num_timesteps);}"]
        post = ["// finally transfer the result back",
                "gpuErrchk(hipMemcpy(hd_data->h_dc_du,hd_data->d_dc_du,NUM_JOINTS*2*NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyDeviceToHost));",
                "gpuErrchk(hipDeviceSynchronize());"]
        self._emit_host("ID_DU", "inverse_dynamics_gradient", "Computes the gradient of inverse dynamics",
                        "template <typename T, bool USE_QDD_FLAG = false, bool USE_COMPRESSED_MEM = false>", True, pre, launches, post, "ID_DU")

    def gen_inverse_dynamics_gradient(self, use_thread_group=False):
        for use_qdd in (True, False):
            doc = "RNEA + analytical gradient core: dc_du = [dc/dq | dc/dqd] at (q, qd%s)" % (", qdd" if use_qdd else ", 0")
            if self.grad_schedule == "recompute":
                self._emit_core("inverse_dynamics_gradient_core" + ("_qdd" if use_qdd else ""), doc + " -- column-serial, recomputing",
                                cores.core_gradient_recompute(self.spec, "id", use_qdd=use_qdd, table=self.grad_table), order="creation")
            else:
                self._emit_core("inverse_dynamics_gradient_core" + ("_qdd" if use_qdd else ""), doc,
                                cores.core_inverse_dynamics_gradient(self.spec, use_qdd))
        if self.emit_inner_api:
            self.gen_inverse_dynamics_gradient_inner(use_thread_group)
        self.gen_inverse_dynamics_gradient_device(use_thread_group, False)
        self.gen_inverse_dynamics_gradient_device(use_thread_group, True)
        self.gen_inverse_dynamics_gradient_kernel(use_thread_group, True)
        self.gen_inverse_dynamics_gradient_kernel(use_thread_group, False)
        n = self.spec.n
        self._emit_split_family("ID_DU", "inverse_dynamics_gradient_kernel", "inverse_dynamics_gradient_core",
                                "Computes the gradient of inverse dynamics", "dc_du", ("q_qd", 2 * n, "stride_q_qd"), True,
                                "s_q_qd, s_q_qd + %d, nullptr, nullptr, nullptr" % n,
                                self._split_builder("id"), None)
        if self.use_pipeline:
            self._emit_pipeline_family("ID_DU", "inverse_dynamics_gradient", "Computes the gradient of inverse dynamics", "dc_du",
                                       ("q_qd", 2 * n, "stride_q_qd"), True)
        else:
            self._emit_no_pipeline("ID_DU", "inverse_dynamics_gradient", "dc_du", ("q_qd", 2 * n, "stride_q_qd"))
        self.gen_inverse_dynamics_gradient_lean_decl()
        self.gen_inverse_dynamics_gradient_host()

    # ------------------------------------------------------------------------------------------
    # forward dynamics gradient (headline)
    # ------------------------------------------------------------------------------------------
    def gen_forward_dynamics_gradient_inner_temp_mem_size(self, use_qdd_Minv_input=False):
        return 0

    def gen_forward_dynamics_gradient_kernel_max_temp_mem_size(self):
        return 0

    def gen_forward_dynamics_gradient_device(self, use_thread_group=False, use_qdd_Minv_input=False):
        notes = ["Uses the fd/du = -Minv*id/du trick as described in Carpentier and Mansrud 'Analytical Derivatives of Rigid Body Dynamics Algorithms'"]
        self.gen_add_func_doc("Computes the gradient of forward dynamics", notes, [], None)
        extra = "const T *s_qdd, const T *s_Minv, " if use_qdd_Minv_input else "const T *s_u, "
        self.gen_add_code_lines(["template <typename T, typename C = typename grid_compute<T>::type>",
                                 "__host__ __device__ __forceinline__",
                                 "void forward_dynamics_gradient_device(T *s_df_du, const T *s_q, const T *s_qd, %sconst robotModel<T> *d_robotModel, const T gravity) {" % extra], True)
        acc = "s_q, s_qd, nullptr, s_qdd, s_Minv" if use_qdd_Minv_input else "s_q, s_qd, s_u, nullptr, nullptr"
        tab = self.io_layout["FD_DU"].get("table", 0)
        if tab:
            self.gen_add_code_line("T s_tab[%d];   // lane-private table of the recomputing core" % tab)
            acc += ", s_tab"
        self.gen_add_code_lines(["(void)d_robotModel;",
                                 "const grid_in_ptrs<T> in = {%s};" % acc,
                                 "grid_out_ptr<T> out = {s_df_du};",
                                 "forward_dynamics_gradient_core%s<T,C>(in, out, gravity);" % ("_qdd_minv" if use_qdd_Minv_input else "")])
        self.gen_add_end_function()

    def gen_forward_dynamics_gradient_kernel(self, use_thread_group=False, use_qdd_Minv_input=False, single_call_timing=False):
        n = self.spec.n
        if use_qdd_Minv_input:
            self._emit_kernel("FD_DU", "forward_dynamics_gradient_kernel", "forward_dynamics_gradient_core_qdd_minv",
                              "Computes the gradient of forward dynamics", "df_du", ("q_qd", 2 * n, "stride_q_qd"),
                              [("qdd", n), ("Minv", n * n)], True, "s_q_qd, s_q_qd + %d, nullptr, s_qdd, s_Minv" % n,
                              chunk=n if self.grad_schedule == "recompute" else None)
        else:
            self._emit_kernel("FD_DU", "forward_dynamics_gradient_kernel", "forward_dynamics_gradient_core",
                              "Computes the gradient of forward dynamics", "df_du", ("q_qd_u", 3 * n, "stride_q_qd_u"),
                              [], True, "s_q_qd_u, s_q_qd_u + %d, s_q_qd_u + %d, nullptr, nullptr" % (n, 2 * n),
                              chunk=n if self.grad_schedule == "recompute" else None)

    def gen_forward_dynamics_gradient_host(self, mode=0):
        def pre(mode):
            lines = ["const int stride_q_qd = 3*NUM_JOINTS;"]
            if mode == "full":
                lines += ["gpuErrchk(hipMemcpyAsync(hd_data->d_q_qd_u,hd_data->h_q_qd_u,stride_q_qd*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[0]));",
                          "if (USE_QDD_MINV_FLAG) {",
                          "    gpuErrchk(hipMemcpyAsync(hd_data->d_qdd,hd_data->h_qdd,NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[1]));",
                          "    gpuErrchk(hipMemcpyAsync(hd_data->d_Minv,hd_data->h_Minv,NUM_JOINTS*NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyHostToDevice,streams[2]));",
                          "}"]
            return lines
        launches = [
            "if (USE_QDD_MINV_FLAG) {forward_dynamics_gradient_kernel<T>@L(hd_data->d_df_du,hd_data->d_q_qd_u,stride_q_qd,hd_data->d_qdd,hd_data->d_Minv,d_robotModel,gravity,num_timesteps);}",
            # the reference-named wrappers serve the reference's callers: where the tile-cooperative kernel is the faster one
            # (FD_DU_COOP_AUTO_MIN_TILES: large robots) they launch it -- same outputs to round-off, blocks/threads then unused
            "else if (FD_DU_LEAN_AUTO_MIN_TILES > 0 && (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE >= FD_DU_LEAN_AUTO_MIN_TILES && "
            "forward_dynamics_gradient_lean_launch<T>(hd_data->d_df_du,hd_data->d_q_qd_u,stride_q_qd,d_robotModel,gravity,num_timesteps,0,@S)) {}",
            "else if (FD_DU_COOP_AUTO_MIN_TILES > 0 && (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE >= FD_DU_COOP_AUTO_MIN_TILES && "
            "forward_dynamics_gradient_coop_launch<T>(hd_data->d_df_du,hd_data->d_q_qd_u,stride_q_qd,d_robotModel,gravity,num_timesteps,0,@S)) {}",
            "else                   {forward_dynamics_gradient_kernel<T>@L(hd_data->d_df_du,hd_data->d_q_qd_u,stride_q_qd,d_robotModel,gravity,num_timesteps);}"]
        post = ["// finally transfer the result back",
                "gpuErrchk(hipMemcpy(hd_data->h_df_du,hd_data->d_df_du,NUM_JOINTS*2*NUM_JOINTS*num_timesteps*sizeof(T),hipMemcpyDeviceToHost));",
                "gpuErrchk(hipDeviceSynchronize());"]
        self._emit_host("FD_DU", "forward_dynamics_gradient", "Computes the gradient of forward dynamics",
                        "template <typename T, bool USE_QDD_MINV_FLAG = false>", True, pre, launches, post, "FD_DU")

    def gen_forward_dynamics_gradient(self, use_thread_group=False):
        if self.grad_schedule == "recompute":
            self._emit_core("forward_dynamics_gradient_core",
                            "Forward-dynamics gradient core: Minv, RNEA(0), qdd, then column-serial dRNEA (recomputing v, a, f per column) and -Minv*dc_du",
                            cores.core_gradient_recompute(self.spec, "fd", table=self.grad_table), order="creation")
            self._emit_core("forward_dynamics_gradient_core_qdd_minv",
                            "Forward-dynamics gradient core with qdd and (upper triangular) Minv supplied -- column-serial, recomputing",
                            cores.core_gradient_recompute(self.spec, "fd", use_qdd_minv=True, table=self.grad_table), order="creation")
        else:
            self._emit_core("forward_dynamics_gradient_core",
                            "Forward-dynamics gradient core (fused): Minv, RNEA(0), qdd, RNEA(qdd), dRNEA, -Minv*dc_du; out = [dqdd/dq | dqdd/dqd]",
                            cores.core_forward_dynamics_gradient(self.spec, False))
            self._emit_core("forward_dynamics_gradient_core_qdd_minv",
                            "Forward-dynamics gradient core with qdd and (upper triangular) Minv supplied",
                            cores.core_forward_dynamics_gradient(self.spec, True))
        self.gen_forward_dynamics_gradient_device(use_thread_group, False)
        self.gen_forward_dynamics_gradient_device(use_thread_group, True)
        self.gen_forward_dynamics_gradient_kernel(use_thread_group, True)
        self.gen_forward_dynamics_gradient_kernel(use_thread_group, False)
        n = self.spec.n
        self._emit_split_family("FD_DU", "forward_dynamics_gradient_kernel", "forward_dynamics_gradient_core",
                                "Computes the gradient of forward dynamics", "df_du", ("q_qd_u", 3 * n, "stride_q_qd_u"), True,
                                "s_q_qd_u, s_q_qd_u + %d, s_q_qd_u + %d, nullptr, nullptr" % (n, 2 * n),
                                self._split_builder("fd"), None)
        if self.use_pipeline:
            self._emit_pipeline_family("FD_DU", "forward_dynamics_gradient", "Computes the gradient of forward dynamics", "df_du",
                                       ("q_qd_u", 3 * n, "stride_q_qd_u"), False)
        else:
            self._emit_no_pipeline("FD_DU", "forward_dynamics_gradient", "df_du", ("q_qd_u", 3 * n, "stride_q_qd_u"))
        # (the host wrappers follow the tile-cooperative kernel: GRiDCodeGenerator.gen_all_code -- they dispatch it where it is the
        # faster kernel)

    # ------------------------------------------------------------------------------------------
    # rollout consumer of the forward-dynamics gradient
    # ------------------------------------------------------------------------------------------
    def gen_forward_dynamics_gradient_rollout(self, use_thread_group=False):
        """A downstream consumer of the forward-dynamics gradient: the reference ships the `_device` tier so that user kernels
        can keep a trajectory on-chip (README.md:26-29, algorithms/_forward_dynamics_gradient.py:59-99) but contains no such
        kernel.  One lane integrates ONE trajectory with semi-implicit Euler for NUM_STEPS steps; q, qd never leave the lane's
        registers between steps; every step writes the next state and the discrete-time linearisation (A, B)."""
        n = self.spec.n
        row = cores.rollout_row_count(self.spec)
        if self.grad_schedule == "recompute":
            bases = []
            tr = cores.core_gradient_recompute(self.spec, "fd", rollout=bases)
            order = "creation"
        else:
            tr, bases = cores.core_rollout_step(self.spec)
            order = None
        self.gen_add_code_line("const int ROLLOUT_ROW_COUNT = %d; // values per configuration and step: [x+ (%d) | A (%d x %d, column-major) | B (%d x %d)]"
                               % (row, 2 * n, 2 * n, 2 * n, 2 * n, n))
        self.gen_add_code_line("const int ROLLOUT_DYNAMIC_SHARED_MEM_COUNT = %d;" % ((self.max_threads // WAVE) * self.lds_per_wave("ROLLOUT")))
        self._emit_core("forward_dynamics_gradient_rollout_step_core",
                        "One semi-implicit Euler step of the forward dynamics and its linearisation: qdd = FD(q, qd, u); qd+ = qd + dt qdd; "
                        "q+ = q + dt qd+; A = dx+/dx, B = dx+/du (x = [q; qd]); out = chunks of %d values, chunk k at row offset BASES[k]" % (2 * n),
                        tr, order=order)
        self.kernel_instances.append("__global__ void @NS::forward_dynamics_gradient_rollout_kernel<T>(T *, const T *, const T *, const T, "
                                     "const @NS::robotModel<T> *, const T, const int, const int);")
        self.gen_add_func_doc("Trajectory rollout with linearisation (consumer of the forward-dynamics gradient)",
                              ["lane-per-trajectory: the state stays in registers across the NUM_STEPS steps",
                               "d_traj is TIME-MAJOR: d_traj[(t*NUM_TIMESTEPS + k)*ROLLOUT_ROW_COUNT + ...] = [x_{t+1} | A_t | B_t] of trajectory k",
                               "d_u_traj is time-major too: d_u_traj[(t*NUM_TIMESTEPS + k)*NUM_JOINTS + j]",
                               "launch with ROLLOUT_DYNAMIC_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS (or grid_lds_bytes for smaller blocks)"],
                              ["d_traj is the output, NUM_STEPS*NUM_TIMESTEPS*ROLLOUT_ROW_COUNT values",
                               "d_x0 holds the initial states [q | qd], %d values per trajectory (dense)" % (2 * n),
                               "d_u_traj holds the input torques", "dt is the integration step",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "gravity is the gravity constant", "NUM_TIMESTEPS is the number of trajectories (the batch)",
                               "NUM_STEPS is the number of integration steps"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(GRID_MAX_THREADS)")
        self.gen_add_code_line("void forward_dynamics_gradient_rollout_kernel(T *d_traj, const T *d_x0, const T *d_u_traj, const T dt, "
                               "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS, const int NUM_STEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "const grid_tile_iter it(NUM_TIMESTEPS);",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % self.lds_per_wave("ROLLOUT"),
            "for (int k0 = it.k0_first; k0 < NUM_TIMESTEPS; k0 += it.k0_step){",
        ])
        self.indent_level += 1
        self.gen_add_code_line("T s_x[%d];" % (2 * n))
        self._emit_load("s_x", "d_x0", 2 * n, str(2 * n))
        self.gen_add_code_line("for (int t = 0; t < NUM_STEPS; t++){", True)
        self.gen_add_code_line("T s_u[%d]; T s_xn[%d];" % (n, 2 * n))
        self._emit_load("s_u", "(d_u_traj + (size_t)t*NUM_TIMESTEPS*%d)" % n, n, str(n))
        self.gen_add_code_line("const grid_in_ptrs<T> in = {s_x, s_x + %d, s_u, nullptr, nullptr, nullptr, dt};" % n)
        self.gen_add_code_line("grid_out_chunks<T,%d,%d,%s> out = {s_wave, d_traj + (size_t)t*NUM_TIMESTEPS*%d, k0, it.lane, it.W, NUM_TIMESTEPS, s_xn};"
                               % (row, 2 * n, ",".join(str(b) for b in bases), row))
        self.gen_add_code_line("forward_dynamics_gradient_rollout_step_core<T,C>(in, out, gravity);")
        self.gen_add_code_line("#pragma unroll")
        self.gen_add_code_line("for (int i = 0; i < %d; i++){s_x[i] = s_xn[i];}" % (2 * n))
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the rollout kernel (asynchronous, on `stream`)", ["illegal launch shapes are replaced by the suggested one"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("void forward_dynamics_gradient_rollout_launch(T *d_traj, const T *d_x0, const T *d_u_traj, const T dt, const robotModel<T> *d_robotModel,")
        self.gen_add_code_line("        const T gravity, const int num_timesteps, const int num_steps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t stream) {", True)
        self.gen_add_code_lines([
            "dim3 blocks, threads; grid_launch_dims(block_dimms, thread_dimms, num_timesteps, &blocks, &threads); grid_fold_launch_z(&blocks, &threads);",
            "const size_t lds_bytes = grid_lds_bytes<T>(threads, %d);" % self.lds_per_wave("ROLLOUT"),
            "forward_dynamics_gradient_rollout_kernel<T><<<blocks,threads,lds_bytes,stream>>>(d_traj,d_x0,d_u_traj,dt,d_robotModel,gravity,num_timesteps,num_steps);",
            "gpuErrchk(hipGetLastError());",
        ])
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # tile-cooperative forward-dynamics gradient: the waves of a block share one tile of 64 configurations
    # ------------------------------------------------------------------------------------------
    ASYM_SPLIT = 8          # the asymmetric split's factor: 8 waves per tile, launched as ONE block of 512 threads (two waves per SIMD of a CU)
    COOP_WAVES = 4
    FLUSH_SLOTS_PER_COLUMN = 400      # instruction-issue slots one gradient column (two flushes of n values) costs a lone wavefront

    def _coop_groups_fused(self, builder, slots):
        """Column groups for the fused (small-robot) cooperative cores, chosen on a timeline model of the block:

            B1 = max(Minv of the producer, qdd-independent work of each consumer)      (coop_phase_costs()[0]; consumers hoist it)
            B2 = B1 + the producer's qdd product
            end = B2 + max over waves of what is left (consumers: the qdd-dependent part; producer: all of its columns' work)

        The producer takes a (possibly empty) suffix of the columns, the consumers contiguous groups of the rest; exhaustive over
        the few candidates, every candidate traced exactly.  Returns [(role, cols)], producer first."""
        import itertools
        n, W = self.spec.n, self.COOP_WAVES
        memo = {}

        def costs(role, cols):
            key = ("producer" if role == "producer" else "consumer", tuple(cols))
            if key not in memo:
                memo[key] = cores.coop_phase_costs(builder(role, list(cols), slots))
            return memo[key]
        p1_0, p2_0 = costs("producer", ())
        best = None
        for kp in range(0, max(1, n - (W - 1)) + 1):
            pcols = tuple(range(n - kp, n))
            rest = n - kp
            if rest < W - 1:
                continue
            pp1, pp2 = costs("producer", pcols)
            left_prod = (pp1 - p1_0) + (pp2 - p2_0)
            for cuts in itertools.combinations(range(1, rest), W - 2):
                bounds = (0,) + cuts + (rest,)
                groups = [tuple(range(bounds[i], bounds[i + 1])) for i in range(W - 1)]
                cc = [costs("consumer", gcols) for gcols in groups]
                end = max([p1_0] + [c[0] for c in cc]) + p2_0 + max([left_prod] + [c[1] for c in cc])
                if best is None or end < best[0]:
                    best = (end, pcols, groups)
        end, pcols, groups = best
        self.coop_model_end = end
        out = [("producer", list(pcols))]
        for i, gcols in enumerate(groups):
            out.append(("consumer_c" if i == 0 else "consumer", list(gcols)))
        return out

    def _coop_groups(self, builder, slots):
        """Column groups of the cooperating waves: contiguous, balanced on the work each wave has left after the second barrier
        (the producer must recompute RNEA there, the consumers only its qdd-dependent part).  The producer takes the LAST
        group (late columns are the cheapest).  Returns [(role, cols)], producer first."""
        n, W = self.spec.n, self.COOP_WAVES
        two = slots.ksplit is not None
        # role of the wave that takes group i (groups in column order): the first group goes to the wave that publishes c, the
        # last to the producer -- with two producers the last two (late columns are the cheapest)
        def role_name(i):
            if i == W - 1:
                return "producer"
            if two and i == W - 2:
                return "producer2"
            if slots.hoist_budget:      # the wave with the heaviest (earliest) columns has the whole idle time to park some of them:
                return "consumer_c" if i == 1 else "consumer"       # the next one computes and publishes c
            return "consumer_c" if i == 0 else "consumer"
        def ops(role, cols):
            # what decides the block's time is the work AFTER the second barrier: every wave leaves it at the same moment (when the
            # producer has published qdd), so the block finishes with the wave that has the most phase-2 instructions
            # (arithmetic + exchange-region reads issued after the barrier)
            tr = builder(role, cols, slots)
            live = tr.live_nodes()
            barriers = [pos for (dst, _), pos in zip(tr.outputs, tr.out_pos) if dst == "barrier"]
            start = barriers[-1]
            arith = ("fma", "mul", "add", "pkfma", "pkmul", "pkadd")
            return self.FLUSH_SLOTS_PER_COLUMN * len(cols) + sum(1 for k in range(start, len(tr.nodes)) if live[k] and (
                tr.nodes[k][0] in arith or (tr.nodes[k][0] == "in" and str(tr.nodes[k][1]).startswith("in.xch_get("))))
        single = {role: [ops(role, [c]) for c in range(n)] for role in ("producer", "consumer")}
        base = {role: min(single[role]) for role in single}
        marg = {role: [x - base[role] for x in single[role]] for role in single}

        def cost(role, b, e):
            r = "producer" if role.startswith("producer") else "consumer"
            return base[r] + sum(marg[r][b:e])
        best = None
        import itertools
        for cuts in itertools.combinations(range(1, n), W - 1):
            bounds = (0,) + cuts + (n,)
            parts = [(bounds[i], bounds[i + 1]) for i in range(W)]
            worst = max(cost(role_name(i), *parts[i]) for i in range(W))
            if best is None or worst < best[0]:
                best = (worst, parts)
        parts = list(best[1])
        # the additive model is rough for large robots (what a column costs depends on its neighbours): hill-climb on EXACT costs
        role_of = (lambda i: role_name(i)) if slots.hoist_budget else (lambda i: ("consumer" if role_name(i) == "consumer_c" else role_name(i)))
        exact = {}

        def cost_exact(i, pr):
            key = (role_of(i), pr)
            if key not in exact:
                exact[key] = ops(role_of(i), list(range(*pr)))
            return exact[key]
        # coordinate descent on the W-1 cut points (moves of one or two columns; a move may first make the maximum worse for a
        # neighbour that a later move relieves, so ties on the maximum are broken by the sum of squares)
        def score(pp):
            cs = [cost_exact(i, pp[i]) for i in range(W)]
            return (max(cs), sum(c * c for c in cs))
        for _ in range(12 if n > 8 else 0):
            cur = score(parts)
            best_move = None
            for cut in range(1, W):
                for step in (-2, -1, 1, 2):
                    pos = parts[cut][0] + step
                    if pos <= parts[cut - 1][0] or pos >= parts[cut][1]:
                        continue
                    trial = list(parts)
                    trial[cut - 1] = (parts[cut - 1][0], pos); trial[cut] = (pos, parts[cut][1])
                    sc = score(trial)
                    if sc < cur and (best_move is None or sc < best_move[0]):
                        best_move = (sc, trial)
            if best_move is None:
                break
            parts = best_move[1]
        order = [W - 1] + ([W - 2] if two else []) + [i for i in range(W - 1) if not (two and i == W - 2)]      # producer(s) first
        return [(role_name(i), list(range(*parts[i]))) for i in order]

    def _coop_prefix_split(self, builder, slots):
        """Column k0 at which two producer waves divide the forward pass of the Minv recursion (columns k < k0 | k >= k0): the one
        that minimises the larger of the two waves' arithmetic up to the second barrier (each wave's trace keeps only the part of
        the backward pass its own columns need)."""
        n = self.spec.n
        arith = ("fma", "mul", "add", "pkfma", "pkmul", "pkadd")

        def between_barriers(role, k0):
            slots.ksplit = k0
            tr = builder(role, [], slots)
            live = tr.live_nodes()
            b = [pos for (dst, _), pos in zip(tr.outputs, tr.out_pos) if dst == "barrier"]
            return sum(1 for k in range(1, b[1]) if live[k] and tr.nodes[k][0] in arith)
        best = None
        for k0 in range(n // 4, n - n // 8):
            worst = max(between_barriers("producer", k0), between_barriers("producer2", k0))
            if best is None or worst < best[0]:
                best = (worst, k0)
        slots.ksplit = None
        return best[1], best[0]

    def gen_forward_dynamics_gradient_coop(self, use_thread_group=False):
        """`forward_dynamics_gradient_kernel_coop`: one block of COOP_WAVES wavefronts per tile of 64 configurations.  The
        column-split kernels repeat the shared prefix (X(q), Minv, RNEA) in every column group; here the waves of a block
        SHARE it: one wave runs the Minv recursion while the others run RNEA, the results cross through LDS, and every wave
        then differentiates its own group of columns.  Large robots also stop holding Minv in registers (it is read from LDS
        where it is used), which is what made their column groups spill."""
        n = self.spec.n
        W = self.COOP_WAVES
        if n < W:
            self.gen_add_code_line("const int FD_DU_COOP_WAVES = 0; // no tile-cooperative kernel for this robot")
            self._emit_no_coop()
            return
        slots = cores.CoopSlots(self.spec)
        rec = (self.grad_schedule == "recompute")
        if rec:
            builder = lambda role, cols, sl: cores.core_gradient_recompute(self.spec, "fd", cols=cols, coop=(role, sl))
            if n > 12:
                # Two producer waves: the serial prefix (backward pass 4.4 k + forward pass and qdd 3.4 k arithmetic instructions for
                # Atlas-30, during which the consumers idle) shrinks by the half of the forward pass the second producer takes.
                # Not in the mixed arithmetic: the two shares of qdd = Minv (u - c) would be rounded to float before they are added.
                if self.precision != "mixed":
                    slots.ksplit, prefix = self._coop_prefix_split(builder, slots)
                else:       # one producer: its whole recursion (up to the second barrier) is the consumers' idle time
                    ptr = builder("producer", [], slots)
                    plive = ptr.live_nodes()
                    pb = [pos for (dst, _), pos in zip(ptr.outputs, ptr.out_pos) if dst == "barrier"]
                    prefix = sum(1 for k in range(1, pb[1]) if plive[k] and ptr.nodes[k][0] in ("fma", "mul", "add", "pkfma", "pkmul", "pkadd"))
                # ... and the consumer waves fill what is left of their idle time with the d/dqd recursions of some of their columns
                # (CoopSlots.hoisted_columns; cores.core_gradient_recompute): cost per column traced here, budget = 90 % of the idle
                # arithmetic (the wave that publishes c has RNEA to do first)
                arith = ("fma", "mul", "add", "pkfma", "pkmul", "pkadd")

                def before_first_barrier(role, cols):
                    tr = builder(role, cols, slots)
                    live = tr.live_nodes()
                    b0 = [pos for (dst, _), pos in zip(tr.outputs, tr.out_pos) if dst == "barrier"][0]
                    return sum(1 for k in range(1, b0) if live[k] and tr.nodes[k][0] in arith)
                rnea_ops = before_first_barrier("consumer_c", [])
                slots.hoist_cost = [1] * n
                slots.hoist_budget = {"consumer": 10 ** 9}
                slots.hoist_cost = [before_first_barrier("consumer", [c]) for c in range(n)]
                slots.hoist_budget = {"consumer": int(prefix), "consumer_c": int(0.9 * max(0, prefix - rnea_ops))}
        else:
            builder = lambda role, cols, sl: cores.core_forward_dynamics_gradient_coop(self.spec, role, cols, sl, hoist=self.coop_hoist)
        groups = self._coop_groups_fused(builder, slots) if (self.coop_hoist and not rec) else self._coop_groups(builder, slots)
        n_out = self.io_layout["FD_DU"]["n_out"]
        piece = min(32, max(16, n))                  # inputs staged in small pieces: small per-wave staging regions (Atlas-30:
                                                     # 4 x 7.5 KB + 495 exchange slots x 256 B = 154 KB of the CU's 160 KB)
        stage = WAVE * max(piece, n)                 # per wave: input pieces, and one gradient column per flush
        xch_off = W * stage
        lds_elems = xch_off + WAVE * slots.count
        if 4 * lds_elems > 160 * 1024:
            self.gen_add_code_line("const int FD_DU_COOP_WAVES = 0; // the exchange region of this robot does not fit the 160 KB of LDS")
            self.note("no tile-cooperative kernel (FD_DU_COOP_WAVES = 0): its exchange region -- the non-zero upper triangle of Minv, %d slots of "
                      "256 B, plus staging -- needs %d KB of the CU's 160 KB of LDS; the forward-dynamics gradient falls back to the column-split / "
                      "unsplit recomputing kernels, which keep Minv in registers" % (slots.count, 4 * lds_elems // 1024))
            self._emit_no_coop()
            return
        self.coop_stats = dict(groups=[(r, list(c)) for (r, c) in groups], slots=slots.count, lds_bytes=4 * lds_elems)
        self.gen_add_code_line("const int FD_DU_COOP_WAVES = %d; // wavefronts per block of the tile-cooperative kernel (block = %d threads, one tile)" % (W, W * WAVE))
        # from how many tiles on the C ABI picks this kernel by itself.  Large robots: with two producer waves and parked d/dqd
        # recursions it beats the column split at every batch size (Atlas-30: 59 vs 64 us for one tile, 71 vs 100 us at K = 16384);
        # without them (mixed arithmetic) only once the chip is full (K = 16384: 134 vs 188 us; K = 4096: 125 vs 106 us).  Small
        # robots: never by itself (iiwa-7: 12.0 vs 11.0 us).
        auto_tiles = (1 if slots.ksplit is not None else 192) if n > 12 else 0
        if self.precision == "fp64":
            auto_tiles = 0          # (the exchange region holds T = float: an all-double build keeps to its all-double kernels by itself)
        self.gen_add_code_line("const int FD_DU_COOP_AUTO_MIN_TILES = %d; // automatic choice of the tile-cooperative kernel from this many tiles on (0: only on request)" % auto_tiles)
        self.gen_add_code_line("const int FD_DU_COOP_SHARED_MEM_COUNT = %d; // dynamic LDS in T elements: %d staging regions + %d exchange slots x 64 lanes"
                               % (lds_elems, W, slots.count))
        names = []
        for w, (role, cols) in enumerate(groups):
            cname = "forward_dynamics_gradient_coop_core_w%d" % w
            tr = builder(role, cols, slots)
            self._emit_core(cname, "Tile-cooperative forward-dynamics gradient, wave %d of %d (%s): columns %s of d/dq and of d/dqd"
                            % (w, W, role, list(cols)), tr, order="creation" if rec else None)
            names.append((cname, cols))
        self.kernel_instances.append("__global__ void @NS::forward_dynamics_gradient_kernel_coop<T>(T *, const T *, const int, "
                                     "const @NS::robotModel<T> *, const T, const int);")
        self.gen_add_func_doc("Computes the gradient of forward dynamics (tile-cooperative: %d wavefronts share each tile of 64 configurations)" % W,
                              ["launch with EXACTLY %d threads per block and FD_DU_COOP_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS" % (W * WAVE),
                               "(use forward_dynamics_gradient_coop_launch); blocks grid-stride over the tiles",
                               "wave roles and column groups: %s" % ["%s:%s" % (r, list(c)) for (r, c) in groups]],
                              ["d_df_du is the output buffer, %d values per configuration" % n_out,
                               "d_q_qd_u is the input buffer, %d values read per configuration" % (3 * n),
                               "stride_q_qd_u is the stride between configurations in d_q_qd_u",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "gravity is the gravity constant", "NUM_TIMESTEPS is the number of configurations"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(%d)" % (W * WAVE))
        self.gen_add_code_line("void forward_dynamics_gradient_kernel_coop(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, "
                               "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "grid_tile_iter it(NUM_TIMESTEPS);              // lane / wave bookkeeping only: the tile loop below is per BLOCK",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % stage,
            "T *s_xch = reinterpret_cast<T *>(s_grid_dyn) + %d;" % xch_off,
            "const int nblocks = grid_num_blocks();",
            "const int bid = grid_block_id();",
            "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it; a wrong shape must not deadlock the barriers)" % (W * WAVE),
            "for (int k0 = bid*GRID_WAVE_SIZE; k0 < NUM_TIMESTEPS; k0 += nblocks*GRID_WAVE_SIZE){",
        ])
        self.indent_level += 1
        self.gen_add_code_line("T s_q_qd_u[%d];" % (3 * n))
        self._emit_load("s_q_qd_u", "d_q_qd_u", 3 * n, "stride_q_qd_u", piece)
        self.gen_add_code_line("const grid_in_coop<T> in = {s_q_qd_u, s_q_qd_u + %d, s_q_qd_u + %d, s_xch, it.lane, "
                               "(unsigned)reinterpret_cast<unsigned long long>(s_xch) + (unsigned)sizeof(T)*it.lane};" % (n, 2 * n))
        self.gen_add_code_line("switch (it.wave_in_block){", True)
        for w, (cname, cols) in enumerate(names):
            if not cols:        # a producer without gradient columns: Minv and qdd only
                self.gen_add_code_line("case %d: {" % w, True)
                self.gen_add_code_line("grid_out_ptr<T> out = {nullptr};     // (this core stores nothing)")
                self.gen_add_code_line("%s<T,C>(in, out, gravity);" % cname)
                self.gen_add_code_line("break;")
                self.gen_add_end_control_flow()
                continue
            len0 = n * len(cols)
            ch = n if rec else self._chunk_for(len0)
            ch = min(ch, max(piece, n)) if 64 * ch > stage else ch
            while len0 % ch != 0:
                ch -= 1
            assert 64 * ch <= stage
            self.gen_add_code_line("case %d: {" % w, True)
            self.gen_add_code_line("grid_out_staged<T,%d,%d,%d,%d,%d,%d> out = {s_wave, d_df_du, k0, it.lane, it.W, NUM_TIMESTEPS};"
                                   % (n_out, 2 * len0, ch, n * cols[0], len0, n * n + n * cols[0]))
            self.gen_add_code_line("%s<T,C>(in, out, gravity);" % cname)
            self.gen_add_code_line("break;")
            self.gen_add_end_control_flow()
        self.gen_add_code_line("default: break;")
        self.gen_add_end_control_flow()
        self.gen_add_code_line("grid_block_sync();     // the exchange region is rewritten by the next tile")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the tile-cooperative forward-dynamics-gradient kernel (asynchronous, on `stream`)",
                              ["tile_blocks <= 0: one block per tile of 64 configurations (capped at 4*SUGGESTED_MAX_BLOCKS)",
                               "returns false when this robot has no tile-cooperative kernel"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_gradient_coop_launch(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, "
                               "const T gravity, const int num_timesteps, int tile_blocks, hipStream_t stream) {", True)
        self.gen_add_code_lines([
            "const size_t lds_bytes = (size_t)FD_DU_COOP_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // > 64 KiB of dynamic LDS must be enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (lds_bytes > 65536 && configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_coop<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));",
            "    configured_device = dev;",
            "}",
            "const int tiles = (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE;",
            "if (tile_blocks <= 0 || tile_blocks > tiles){tile_blocks = tiles;}",
            "if (tile_blocks > 4*SUGGESTED_MAX_BLOCKS){tile_blocks = 4*SUGGESTED_MAX_BLOCKS;}",
            "forward_dynamics_gradient_kernel_coop<T><<<dim3(tile_blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(d_df_du,d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);" % (W * WAVE),
            "gpuErrchk(hipGetLastError());",
            "return true;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the tile-cooperative kernel", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_gradient_coop_attributes(hipFuncAttributes *attr) {", True)
        self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_coop<T>))); return true;")
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # wave-per-configuration forward-dynamics gradient: the lanes of ONE wavefront share a configuration
    # ------------------------------------------------------------------------------------------
    def _wave_spread_lines(self, kernel):
        """Launcher lines of a wave-per-configuration kernel compiled for two waves per SIMD: the dynamic LDS request is padded so that a
        CU's 160 KB admit only as many blocks as an EVEN spread over the 256 CUs needs (ceil(blocks / 256)).  Without it the dispatcher
        packs twice the blocks on some CUs and leaves others idle as soon as the registers allow it: Atlas-30 dFD at K = 512 25.6 us
        against 20.9 us for the 295-register build, which the register file itself limits to two blocks per CU
        (profiles/r04/wave_occupancy.txt)."""
        if self.wave_occupancy <= 1:
            return []
        return ["{   // spread: at most ceil(blocks / CUs) blocks fit one CU's LDS",
                "    const int per_cu = (blocks + GRID_NUM_CUS - 1)/GRID_NUM_CUS;",
                "    const size_t spread = (size_t)GRID_LDS_PER_CU/(size_t)(per_cu < 1 ? 1 : per_cu);",
                "    if (spread > lds_bytes){lds_bytes = spread;}",
                "}"]

    def _wave_occupancy_arg(self):
        """Second __launch_bounds__ argument of the wave-per-configuration kernels (waves per SIMD the compiler must leave room for).
        Large robots: 2 -- a block is one configuration, at 295 registers 512 blocks were resident and the time per launch doubled
        from K = 1024 on; at <= 256 registers 1024 are (experimental wave_occupancy; measured in profiles/r04)."""
        occ = int(self.wave_occupancy)
        return ", %d" % occ if occ > 1 else ""

    def _emit_no_wave(self):
        self.gen_add_code_line("const int FD_DU_WAVE_WAVES = 0; // no wave-per-configuration kernel for this robot")
        self.gen_add_code_line("const int FD_DU_WAVE_AUTO_MAX_K = 0;")
        self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_wave_launch(T *, const T *, const int, const robotModel<T> *, const T, const int, int, hipStream_t) {return false;}",
                                 "template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_wave_attributes(hipFuncAttributes *) {return false;}", ""])

    def gen_forward_dynamics_gradient_wave(self, use_thread_group=False):
        """`forward_dynamics_gradient_kernel_wave`: one BLOCK per configuration, one wavefront per group of base-rooted trees; the
        lanes of a wave are the gradient columns (and the Minv columns) of its group -- the small-batch path (SURVEY.md section
        8(f) rank 2): the reference's block-per-configuration mapping (GRiDCodeGenerator.py:72-83, helpers/_code_generation_helpers.py:
        41-55) redone for 64-wide wavefronts, with cross-lane traffic through v_readlane broadcasts and wave-local LDS instead of
        __syncthreads."""
        n = self.spec.n
        groups = wave.wave_groups(self.spec) if self.precision != "fp64" else None
        if not groups:
            if self.precision != "fp64":
                self.note("no wave-per-configuration kernels (FD_DU_WAVE_WAVES = 0): a wave holds the 2m gradient columns of its group of "
                          "base-rooted trees on its 64 lanes, and a group of this robot has more than 32 joints; small batches run the "
                          "lane-per-configuration kernels")
            self._emit_no_wave()
            return
        W = len(groups)
        per_wave = 0
        layout = []
        for (first, m) in groups:
            ut = wave.WaveTable(m).count
            utab_elems = 2 * ut + 2 * WAVE             # table + the scratch words the lanes other than 0 write (own lanes, a helper wave's lanes)
            mat_elems = 2 * 32 * m + WAVE              # published matrix (row stride 32) + scratch of the lanes >= m
            out_elems = WAVE * n                       # output image [64 columns][n rows]
            layout.append((ut, utab_elems, mat_elems, out_elems))
            per_wave = max(per_wave, utab_elems + mat_elems + out_elems)
        self.wave_stats = dict(groups=list(groups), lds_bytes=4 * per_wave * W)
        self.wave_layout, self.wave_per_wave_elems = layout, per_wave
        self.gen_add_code_line("const int FD_DU_WAVE_WAVES = %d; // wavefronts per block of the wave-per-configuration kernel: one block per configuration, "
                               "joint groups %s" % (W, [list(range(f, f + m)) for (f, m) in groups]))
        # batch sizes up to which the C ABI picks this kernel by itself: large robots while the batch leaves most of the chip idle
        # (one wave per group and configuration; measured against the tile-cooperative kernel in profiles/r03/latency_*.txt);
        # small robots only on request (iiwa-7: the 7-way column split is faster)
        self.gen_add_code_line("const int FD_DU_WAVE_AUTO_MAX_K = %d; // automatic choice of the wave-per-configuration kernel up to this batch size (0: only on request)"
                               % (self.wave_auto_max_k if n > 12 else 0))
        self.gen_add_code_line("const int FD_DU_WAVE_SHARED_MEM_COUNT = %d; // dynamic LDS of a block in T elements (%d per wave: uniform table, published Minv, output image)"
                               % (per_wave * W, per_wave))
        if self.wave_occupancy > 1:
            self.gen_add_code_line("const int GRID_NUM_CUS = 256; const int GRID_LDS_PER_CU = 160*1024; // MI355X: what the wave launchers spread their blocks over")
        roles = wave.wave_roles(self.spec, groups)                 # {helper wave: helped wave}
        helped_by = {hd: hr for hr, hd in roles.items()}
        self.wave_stats["roles"] = dict(roles)
        names = []
        for w, (first, m) in enumerate(groups):
            # spatial inertias of the group's joints, entry-major [21][m]: lane l reads entry e of joint l mod m (in.lane_I)
            upper = [(r, c) for r in range(6) for c in range(r, 6)]
            vals = [repr(float(self.spec.Imats[first + k][r, c])) for (r, c) in upper for k in range(m)]
            self.gen_add_code_line("static __device__ const float FD_DU_WAVE_INERTIA_W%d[%d] = {%s};" % (w, 21 * m, ", ".join(vals)))
        for w, (first, m) in enumerate(groups):
            cname = "forward_dynamics_gradient_wave_core_w%d" % w
            kw = {}
            role = ""
            if w in roles:
                f2, m2 = groups[roles[w]]
                kw["helper_for"] = SubForest(self.spec, f2, m2)
                role = "; first runs the first RNEA pass of wave %d's joints %d..%d (that wave is busy with its Minv recursion)" % (roles[w], f2, f2 + m2 - 1)
            if w in helped_by:
                kw["helped"] = True
                role = "; its first RNEA pass is run by wave %d" % helped_by[w]
            tr = wave.core_forward_dynamics_gradient_wave(SubForest(self.spec, first, m), barriers=bool(roles), **kw)
            self._emit_core(cname, "Wave-per-configuration forward-dynamics gradient, joints %d..%d: lane l < %d is column l of d/dq, lane %d + l of d/dqd%s"
                            % (first, first + m - 1, m, m, role), tr, order="creation")
            names.append(cname)
        self.kernel_instances.append("__global__ void @NS::forward_dynamics_gradient_kernel_wave<T>(T *, const T *, const int, "
                                     "const @NS::robotModel<T> *, const T, const int);")
        self.gen_add_func_doc("Computes the gradient of forward dynamics (wave-per-configuration: the 64 lanes of a wavefront share ONE configuration)",
                              ["launch with EXACTLY %d threads per block and FD_DU_WAVE_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS" % (W * WAVE),
                               "(use forward_dynamics_gradient_wave_launch); block b computes configuration b, b + gridDim, ...",
                               "wave w of a block owns the joints %s (base-rooted trees do not interact)" % [list(range(f, f + m)) for (f, m) in groups]],
                              ["d_df_du is the output buffer, %d values per configuration" % (2 * n * n),
                               "d_q_qd_u is the input buffer, %d values read per configuration" % (3 * n),
                               "stride_q_qd_u is the stride between configurations in d_q_qd_u",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "gravity is the gravity constant", "NUM_TIMESTEPS is the number of configurations"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(%d%s)" % (W * WAVE, self._wave_occupancy_arg()))
        self.gen_add_code_line("void forward_dynamics_gradient_kernel_wave(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, "
                               "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "(void)d_robotModel;",
            "const int tid = grid_thread_id();",
            "const int lane = tid & (GRID_WAVE_SIZE - 1);",
            "const int wave = __builtin_amdgcn_readfirstlane(tid / GRID_WAVE_SIZE);",
            "const int nblocks = grid_num_blocks();",
            "const int bid = grid_block_id();",
            "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it)" % (W * WAVE),
            "T *s_w = reinterpret_cast<T *>(s_grid_dyn) + wave*%d;" % per_wave,
            "switch (wave){", ], True)
        for w, (first, m) in enumerate(groups):
            ut, utab_elems, mat_elems, out_elems = layout[w]
            self.gen_add_code_line("case %d: {" % w, True)
            self.gen_add_code_lines([
                "const int kcol = lane %% %d;" % m,
                "T *s_utab = s_w; T *s_mat = s_w + %d; T *s_out = s_w + %d;" % (utab_elems, utab_elems + mat_elems),
                "T *wput = (lane == 0) ? s_utab : (s_utab + %d + lane);     // only lane 0 writes the table proper" % ut,
                "T *mput = (lane < %d) ? (s_mat + lane) : (s_mat + %d + lane);   // only the lanes that own a column publish it" % (m, 32 * m),
                "grid_out_wave<T,%d,%d,%d> out = {s_out, lane};" % (n, first, m),
                "out.clear();",
                "for (int k = bid; k < NUM_TIMESTEPS; k += nblocks){", ], True)
            self.gen_add_code_lines([
                "const T *row = d_q_qd_u + (size_t)k*stride_q_qd_u;",
                "grid_in_wave<T> in = {row[%d + kcol], row[%d + kcol], row[%d + kcol], wput, mput, lane, kcol, %d, FD_DU_WAVE_INERTIA_W%d,"
                % (first, n + first, 2 * n + first, m, w),
                "                      (unsigned)reinterpret_cast<unsigned long long>(s_utab), (unsigned)reinterpret_cast<unsigned long long>(s_mat)};",
            ])
            if w in roles:
                hw = roles[w]
                f2, m2 = groups[hw]
                ut2 = layout[hw][0]
                self.gen_add_code_lines([
                    "{   // helper role: the first RNEA pass of wave %d's joints goes into THAT wave's table" % hw,
                    "    T *s_utab2 = reinterpret_cast<T *>(s_grid_dyn) + %d*%d;" % (hw, per_wave),
                    "    const int kcol2 = lane %% %d;" % m2,
                    "    in.q2_ = row[%d + kcol2]; in.qd2_ = row[%d + kcol2]; in.kcol2_ = kcol2; in.m2_ = %d; in.inertia2_ = FD_DU_WAVE_INERTIA_W%d;" % (f2, n + f2, m2, hw),
                    "    in.wput2_ = (lane == 0) ? s_utab2 : (s_utab2 + %d + %d + lane);     // (scratch words BEHIND the ones that wave's own lanes use)" % (ut2, WAVE),
                    "    in.utab2_ = (unsigned)reinterpret_cast<unsigned long long>(s_utab2);",
                    "}",
                ])
            self.gen_add_code_lines([
                "%s<T,C>(in, out, gravity);" % names[w],
                "out.flush(d_df_du + (size_t)k*%d);" % (2 * n * n),
            ])
            if roles:
                self.gen_add_code_line("grid_block_sync();     // the tables are rewritten for the next configuration")
            self.gen_add_end_control_flow()
            self.gen_add_code_line("break;")
            self.gen_add_end_control_flow()
        self.gen_add_code_line("default: break;")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the wave-per-configuration forward-dynamics-gradient kernel (asynchronous, on `stream`)",
                              ["blocks <= 0: one block per configuration (capped at 8*SUGGESTED_MAX_BLOCKS; larger batches stride)",
                               "returns false when this robot has no such kernel"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_gradient_wave_launch(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, "
                               "const T gravity, const int num_timesteps, int blocks, hipStream_t stream) {", True)
        self.gen_add_code_lines([
        ] + ([
            "size_t lds_bytes = (size_t)FD_DU_WAVE_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // the padded request (spread) exceeds 64 KiB: enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_wave<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes > GRID_LDS_PER_CU ? lds_bytes : (size_t)GRID_LDS_PER_CU)));",
            "    configured_device = dev;",
            "}",
        ] if self.wave_occupancy > 1 else [
            "const size_t lds_bytes = (size_t)FD_DU_WAVE_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // > 64 KiB of dynamic LDS must be enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (lds_bytes > 65536 && configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_wave<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));",
            "    configured_device = dev;",
            "}",
        ]) + [
            "if (blocks <= 0 || blocks > num_timesteps){blocks = num_timesteps;}",
            "if (blocks > 8*SUGGESTED_MAX_BLOCKS){blocks = 8*SUGGESTED_MAX_BLOCKS;}",
        ] + self._wave_spread_lines("forward_dynamics_gradient_kernel_wave") + [
            "forward_dynamics_gradient_kernel_wave<T><<<dim3(blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(d_df_du,d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);" % (W * WAVE),
            "gpuErrchk(hipGetLastError());",
            "return true;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the wave-per-configuration kernel", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_gradient_wave_attributes(hipFuncAttributes *attr) {", True)
        self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_wave<T>))); return true;")
        self.gen_add_end_function()

    # the other four algorithms on the same wave-per-configuration phases (emit/wave.py: kind): the small-batch path of every kernel
    WAVE_OTHERS = [
        # alg, kind, kernel base, doc, output name, outputs per configuration (as a function of n), inputs
        ("ID", "id", "inverse_dynamics", "Compute the RNEA (Recursive Newton-Euler Algorithm)", "c", lambda n: n, ("q_qd", True, False, True)),
        ("MINV", "minv", "direct_minv", "Compute the inverse of the mass matrix", "Minv", lambda n: n * n, ("q", False, False, False)),
        ("FD", "fd", "forward_dynamics", "Computes forward dynamics", "qdd", lambda n: n, ("q_qd_u", True, True, False)),
        ("ID_DU", "id_du", "inverse_dynamics_gradient", "Computes the gradient of inverse dynamics", "dc_du", lambda n: 2 * n * n, ("q_qd", True, False, True)),
    ]

    def gen_wave_kernels_other(self, use_thread_group=False):
        """`<algorithm>_kernel_wave` for RNEA, Minv, forward dynamics and the RNEA gradient: one block per configuration, one wavefront
        per group of base-rooted trees, the phases of the forward-dynamics-gradient wave kernel that the algorithm needs.  The reference
        runs all five algorithms block-per-configuration (its `_inner` family); these are their wave64 counterparts for batches that
        leave the chip mostly idle."""
        n = self.spec.n
        groups = wave.wave_groups(self.spec) if self.precision != "fp64" else None
        for (alg, kind, base, doc, out_name, n_out_f, (in_name, has_qd, has_u, has_qdd)) in self.WAVE_OTHERS:
            launch_sig = ("bool %s_wave_launch(T *d_%s, const T *d_%s, const int stride_%s, %sconst robotModel<T> *d_robotModel, %sconst int num_timesteps, int blocks, hipStream_t stream)"
                          % (base, out_name, in_name, in_name, "const T *d_qdd, " if has_qdd else "", "const T gravity, " if kind != "minv" else ""))
            if not groups:
                self.gen_add_code_line("const int %s_WAVE_AUTO_MAX_K = 0; // no wave-per-configuration kernel for this robot / arithmetic" % alg)
                self.gen_add_code_lines(["template <typename T>", "__host__ inline", launch_sig.replace(" *d_%s" % out_name, " *").replace("T *d_", "T *") + " {return false;}", ""])
                continue
            W = len(groups)
            per_wave = self.wave_per_wave_elems
            n_out = n_out_f(n)
            auto = self.wave_auto_other.get(alg, 0)
            self.gen_add_code_line("const int %s_WAVE_AUTO_MAX_K = %d; // automatic choice of %s_kernel_wave up to this batch size (0: only on request)" % (alg, auto, base))
            names = []
            for w, (first, m) in enumerate(groups):
                cname = "%s_wave_core_w%d" % (base, w)
                tr = wave.core_forward_dynamics_gradient_wave(SubForest(self.spec, first, m), kind=kind)
                self._emit_core(cname, "%s, wave-per-configuration, joints %d..%d" % (doc, first, first + m - 1), tr, order="creation")
                names.append(cname)
            grav = kind != "minv"
            sig = "void %s_kernel_wave(T *d_%s, const T *d_%s, const int stride_%s, %sconst robotModel<T> *d_robotModel, %sconst int NUM_TIMESTEPS)" % (
                base, out_name, in_name, in_name, "const T *d_qdd, " if has_qdd else "", "const T gravity, " if grav else "")
            self.kernel_instances.append("__global__ void @NS::%s_kernel_wave<T>(%s);" % (
                base, ", ".join(["T *", "const T *", "const int"] + (["const T *"] if has_qdd else []) + ["const @NS::robotModel<T> *"]
                                + (["const T"] if grav else []) + ["const int"])))
            self.gen_add_func_doc(doc + " (wave-per-configuration: the 64 lanes of a wavefront share ONE configuration)",
                                  ["launch with EXACTLY %d threads per block and FD_DU_WAVE_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS (use %s_wave_launch)" % (W * WAVE, base),
                                   "block b computes configuration b, b + gridDim, ...; wave w of a block owns the joints %s" % [list(range(f, f + m)) for (f, m) in groups]]
                                  + (["d_qdd may be nullptr (qdd = 0)"] if has_qdd else []),
                                  ["d_%s is the output buffer, %d values per configuration" % (out_name, n_out),
                                   "d_%s is the input buffer; stride_%s is the stride between configurations in it" % (in_name, in_name)], None)
            self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
            self.gen_add_code_line("__global__ __launch_bounds__(%d%s)" % (W * WAVE, self._wave_occupancy_arg()))
            self.gen_add_code_line(sig + " {", True)
            self.gen_add_code_lines([
                "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
                "(void)d_robotModel;",
                "const int tid = grid_thread_id();",
                "const int lane = tid & (GRID_WAVE_SIZE - 1);",
                "const int wave = __builtin_amdgcn_readfirstlane(tid / GRID_WAVE_SIZE);",
                "const int nblocks = grid_num_blocks();",
                "const int bid = grid_block_id();",
                "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it)" % (W * WAVE),
                "T *s_w = reinterpret_cast<T *>(s_grid_dyn) + wave*%d;" % per_wave,
                "switch (wave){", ], True)
            for w, (first, m) in enumerate(groups):
                ut, utab_elems, mat_elems, out_elems = self.wave_layout[w]
                self.gen_add_code_line("case %d: {" % w, True)
                self.gen_add_code_lines([
                    "const int kcol = lane %% %d;" % m,
                    "T *s_utab = s_w; T *s_mat = s_w + %d; T *s_out = s_w + %d;" % (utab_elems, utab_elems + mat_elems),
                    "T *wput = (lane == 0) ? s_utab : (s_utab + %d + lane);" % ut,
                    "T *mput = (lane < %d) ? (s_mat + lane) : (s_mat + %d + lane);" % (m, 32 * m),
                ])
                if kind in ("minv", "id_du"):
                    self.gen_add_code_line("grid_out_wave<T,%d,%d,%d,%d> out = {s_out, lane};" % (n, first, m, 1 if kind == "minv" else 2))
                    self.gen_add_code_line("out.clear();")
                else:
                    self.gen_add_code_line("(void)s_out;")
                self.gen_add_code_line("for (int k = bid; k < NUM_TIMESTEPS; k += nblocks){", True)
                q_expr = "row[%d + kcol]" % first
                qd_expr = ("row[%d + kcol]" % (n + first)) if has_qd else "static_cast<T>(0)"
                u_expr = ("row[%d + kcol]" % (2 * n + first)) if has_u else "static_cast<T>(0)"
                qdd_expr = ("(d_qdd != nullptr ? d_qdd[(size_t)k*%d + %d + kcol] : static_cast<T>(0))" % (n, first)) if has_qdd else "static_cast<T>(0)"
                self.gen_add_code_lines([
                    "const T *row = d_%s + (size_t)k*stride_%s;" % (in_name, in_name),
                    "const grid_in_wave<T> in = {%s, %s, %s, wput, mput, lane, kcol, %d, FD_DU_WAVE_INERTIA_W%d," % (q_expr, qd_expr, u_expr, m, w),
                    "                            (unsigned)reinterpret_cast<unsigned long long>(s_utab), (unsigned)reinterpret_cast<unsigned long long>(s_mat), %s};" % qdd_expr,
                ])
                g_expr = "gravity" if grav else "static_cast<T>(0)"
                if kind == "id":
                    self.gen_add_code_line("grid_out_wave_vec<T,%d> out = {d_%s + (size_t)k*%d};" % (first, out_name, n_out))
                elif kind == "fd":
                    self.gen_add_code_line("grid_out_wave_lane<T,%d> out = {d_%s + (size_t)k*%d, kcol};" % (first, out_name, n_out))
                self.gen_add_code_line("%s<T,C>(in, out, %s);" % (names[w], g_expr))
                if kind in ("minv", "id_du"):
                    self.gen_add_code_line("out.flush(d_%s + (size_t)k*%d);" % (out_name, n_out))
                self.gen_add_end_control_flow()
                self.gen_add_code_line("break;")
                self.gen_add_end_control_flow()
            self.gen_add_code_line("default: break;")
            self.gen_add_end_control_flow()
            self.gen_add_end_function()
            self.gen_add_func_doc("Launch %s_kernel_wave (asynchronous, on `stream`)" % base,
                                  ["blocks <= 0: one block per configuration (capped at 8*SUGGESTED_MAX_BLOCKS; larger batches stride)",
                                   "returns false when this robot has no such kernel"], [], None)
            self.gen_add_code_line("template <typename T>")
            self.gen_add_code_line("__host__ inline")
            self.gen_add_code_line(launch_sig + " {", True)
            args = ["d_" + out_name, "d_" + in_name, "stride_" + in_name] + (["d_qdd"] if has_qdd else []) + ["d_robotModel"] + (["gravity"] if grav else []) + ["num_timesteps"]
            self.gen_add_code_lines([
            ] + ([
                "size_t lds_bytes = (size_t)FD_DU_WAVE_SHARED_MEM_COUNT*sizeof(T);",
                "static thread_local int configured_device = -1;",
                "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
                "if (configured_device != dev){gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&%s_kernel_wave<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes > GRID_LDS_PER_CU ? lds_bytes : (size_t)GRID_LDS_PER_CU))); configured_device = dev;}" % base,
            ] if self.wave_occupancy > 1 else [
                "const size_t lds_bytes = (size_t)FD_DU_WAVE_SHARED_MEM_COUNT*sizeof(T);",
                "if (lds_bytes > 65536){gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&%s_kernel_wave<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));}" % base,
            ]) + [
                "if (blocks <= 0 || blocks > num_timesteps){blocks = num_timesteps;}",
                "if (blocks > 8*SUGGESTED_MAX_BLOCKS){blocks = 8*SUGGESTED_MAX_BLOCKS;}",
            ] + self._wave_spread_lines("%s_kernel_wave" % base) + [
                "%s_kernel_wave<T><<<dim3(blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(%s);" % (base, W * WAVE, ",".join(args)),
                "gpuErrchk(hipGetLastError());",
                "return true;",
            ])
            self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the wave-per-configuration kernel of algorithm 0..3 (ID, MINV, FD, ID_DU)", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool wave_attributes(int alg, hipFuncAttributes *attr) {", True)
        if not groups:
            self.gen_add_code_line("(void)alg; (void)attr; return false;")
        else:
            self.gen_add_code_line("const void *f = nullptr;")
            for a, (_alg, _kind, base, *_rest) in enumerate(self.WAVE_OTHERS):
                self.gen_add_code_line("if (alg == %d){f = reinterpret_cast<const void *>(&%s_kernel_wave<T>);}" % (a, base))
            self.gen_add_code_line("if (f == nullptr){return false;}")
            self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, f)); return true;")
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # register-lean tile-cooperative forward-dynamics gradient: EIGHT wavefronts share one tile, two per SIMD
    # ------------------------------------------------------------------------------------------
    def _lean_prepare(self):
        """(slots, plan, stage, lds_elems) of the register-lean kernel, or None when this robot / arithmetic has none (cached: the
        declarations are emitted ahead of the host wrappers, the kernel itself last)."""
        if hasattr(self, "_lean_cache"):
            return self._lean_cache
        n = self.spec.n
        W = cores.LEAN_WAVES
        self._lean_cache = None
        # fp32 only.  The cores accept the mixed arithmetic (Minv passes and qdd rows in double inside the waves, float across LDS;
        # experimental lean_mixed=True builds it into the mixed library, on request: grid_set_coop mode 3), but what crosses LDS as
        # float gives the double recursion's accuracy away: Atlas-30 dFD 1.7-8.1e-6 against 0.5-1.1e-6 for the 4-wave mixed kernel and
        # 3.0e-6-1.1e-5 for this kernel in fp32, at 50.9 us against 44.4 (profiles/r04/mixed_lean_report.txt)
        if (n < self.lean_min_joints or (self.precision != "fp32" and not (self.precision == "mixed" and self.lean_mixed))
                or (self.grad_schedule != "recompute" and n > 12)):
            return None
        # (the LDS need is known before the plan, which costs ~2 n traced cores: a robot whose exchange region cannot fit skips it)
        probe_slots = cores.CoopSlots(self.spec)
        probe_slots.enable_lean(self.spec, umc=self.lean_plan_options.get("umc", True))
        if 4 * (W * WAVE * (34 if self.lean_plan_options.get("aligned_flush", True) else n) + WAVE * probe_slots.count) > 160 * 1024:
            self.note("no register-lean 8-wave tile-cooperative kernel (FD_DU_LEAN_WAVES = 0): exchange region + input table + 8 staging "
                      "regions need %d KB of the CU's 160 KB of LDS" % ((W * WAVE * 34 + WAVE * probe_slots.count) * 4 // 1024))
            return None
        slots, plan = cores.lean_plan(self.spec, W, **self.lean_plan_options)
        if self.lean_probe == "prefix":             # (experiment: phases 0-3 only -- what a tile costs before its first gradient column)
            for (role, items) in plan:
                role.hoist = []
            plan = [(role, []) for (role, items) in plan]
        elif self.lean_probe == "older":            # (experiment: only the waves dispatched first keep their columns)
            plan = [(role, items if w < W // 2 else []) for w, (role, items) in enumerate(plan)]
        elif self.lean_probe == "younger":
            plan = [(role, items if w >= W // 2 else []) for w, (role, items) in enumerate(plan)]
        # per wave: one gradient half-column per flush (or one piece of at most 32 values at a pitch of 34: AlignedPieces); before
        # the first flush the staging regions park U, 1/D of the Minv recursion (7 n words per lane) and u - c (n more, lean_umc)
        stage = WAVE * (34 if getattr(slots, "aligned_flush", False) else n)
        lds_elems = W * stage + WAVE * slots.count
        if 4 * lds_elems > 160 * 1024 or (8 if slots.lean_umc else 7) * n > W * (stage // WAVE):
            self.note("no register-lean 8-wave tile-cooperative kernel (FD_DU_LEAN_WAVES = 0): exchange region + input table + 8 staging "
                      "regions need %d KB of the CU's 160 KB of LDS" % (4 * lds_elems // 1024))
            return None
        self._lean_cache = (slots, plan, stage, lds_elems)
        return self._lean_cache

    def gen_forward_dynamics_gradient_lean_decl(self):
        """Constants and forward declarations of the register-lean kernel's launcher, ahead of the reference-named host wrappers (which
        dispatch it where it is the fastest kernel); the kernel itself is emitted last (gen_forward_dynamics_gradient_lean)."""
        prep = self._lean_prepare()
        W = cores.LEAN_WAVES
        if prep is None:
            self.gen_add_code_line("const int FD_DU_LEAN_WAVES = 0; // no register-lean tile-cooperative kernel for this robot / arithmetic")
            self.gen_add_code_line("const int FD_DU_LEAN_AUTO_MIN_TILES = 0;")
            self.gen_add_code_line("const int FD_DU_LEAN_WAVE_MAX_K = 0;")
        else:
            slots, plan, stage, lds_elems = prep
            self.gen_add_code_line("const int FD_DU_LEAN_WAVES = %d; // wavefronts per block of the register-lean tile-cooperative kernel (block = %d threads, one tile)" % (W, W * WAVE))
            # measured against the 4-wave kernel (profiles/r04/lean_sweep.txt, Atlas-30, us per launch, 4 waves -> 8 waves): K = 64 54.3 -> 40.0,
            # 4096 60.8 -> 47.9, 16384 67.0 -> 52.8, 32768 130.6 -> 106.7, 65536 306.9 -> 290.8, 131072 624 -> 568: every batch size
            # mixed arithmetic: the Minv passes and the qdd rows run in double inside the waves, but what crosses LDS is float -- faster
            # than the 4-wave kernel and less accurate than it: there only on request (grid_set_coop mode 3)
            self.gen_add_code_line("const int FD_DU_LEAN_AUTO_MIN_TILES = %d; // automatic choice of the register-lean kernel from this many tiles on (0: only on request)"
                                   % (self.lean_auto_min_tiles if self.precision == "fp32" else 0))
            self.gen_add_code_line("const int FD_DU_LEAN_SHARED_MEM_COUNT = %d; // dynamic LDS in T elements: %d staging regions of %d + %d exchange slots x 64 lanes"
                                   % (lds_elems, W, stage, slots.count))
            # where this kernel overtakes the wave-per-configuration kernel (one block per configuration, 512 resident at a time): Atlas-30
            # wave 20.5 us at K = 512, 39.7 at 1024 against 34.9 / 35.3 us here (profiles/r04/latency_all_atlas30_fp32.txt)
            self.gen_add_code_line("const int FD_DU_LEAN_WAVE_MAX_K = %d; // with this kernel in the library the wave-per-configuration kernel is chosen automatically only up to this batch size (0: FD_DU_WAVE_AUTO_MAX_K alone decides)" % (self.lean_wave_max_k if self.precision == "fp32" else 0))
        self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_lean_launch(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, "
                                 "const T gravity, const int num_timesteps, int tile_blocks, hipStream_t stream);", ""])

    def _emit_no_lean(self):
        self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_lean_launch(T *, const T *, const int, const robotModel<T> *, const T, const int, int, hipStream_t) {return false;}",
                                 "template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_lean_attributes(hipFuncAttributes *) {return false;}", ""])

    def gen_forward_dynamics_gradient_lean(self, use_thread_group=False):
        """`forward_dynamics_gradient_kernel_coop8`: one block of EIGHT wavefronts per tile of 64 configurations -- two per SIMD, at most
        256 registers each -- for large robots in the fp32 arithmetic.  The 4-wave kernel (`..._kernel_coop`) gives every wave a
        SIMD of its own and 464 registers; a lone wave issues one vector instruction per 4 cycles on a SIMD that could take two, and
        39 % of its cycles are waits nobody fills.  Here the cores are built to need HALF a SIMD's registers (cores.CoopSlots.enable_lean,
        cores.lean_plan, alg.minv_backward_lean / minv_forward_lean) so that a second wave fills those slots.  The reference spreads a
        configuration over a 512-thread block (GRiDCodeGenerator.py:72-83; algorithms/_inverse_dynamics_gradient.py:199-246,501-540:
        threads = gradient columns); this is the wave64 counterpart with lanes = configurations and waves = (column half) work items."""
        n = self.spec.n
        W = cores.LEAN_WAVES
        prep = self._lean_prepare()
        if prep is None:
            self._emit_no_lean()
            return
        slots, plan, stage, lds_elems = prep
        piece = n
        xch_off = W * stage
        self.lean_stats = dict(plan=[(repr(r), list(items)) for (r, items) in plan], slots=slots.count, lds_bytes=4 * lds_elems, model=dict(slots.lean_model))
        names = []
        for w, (role, items) in enumerate(plan):
            cname = "forward_dynamics_gradient_lean_core_w%d" % w
            tr = cores.core_gradient_recompute(self.spec, "fd", cols=items, coop=(role, slots))
            self._emit_core(cname, "Register-lean tile-cooperative forward-dynamics gradient, wave %d of %d: %r; gradient half-columns (column, 0 = d/dq | 1 = d/dqd) %s"
                            % (w, W, role, list(items)), tr, order="creation", read_ahead=self.lean_read_ahead)
            names.append((cname, list(tr.run_bases), any(isinstance(d, str) and d.startswith("flush:") for (d, _) in tr.outputs)))
        n_out = self.io_layout["FD_DU"]["n_out"]
        self.kernel_instances.append("__global__ void @NS::forward_dynamics_gradient_kernel_coop8<T>(T *, const T *, const int, "
                                     "const @NS::robotModel<T> *, const T, const int);")
        self.gen_add_func_doc("Computes the gradient of forward dynamics (register-lean tile-cooperative: %d wavefronts, two per SIMD, share each tile of 64 configurations)" % W,
                              ["launch with EXACTLY %d threads per block and FD_DU_LEAN_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS" % (W * WAVE),
                               "(use forward_dynamics_gradient_lean_launch); blocks grid-stride over the tiles",
                               "LDS: [%d staging regions | exchange region]; before the second barrier the staging regions park U and 1/D of the Minv recursion" % W],
                              ["d_df_du is the output buffer, %d values per configuration" % n_out,
                               "d_q_qd_u is the input buffer, %d values read per configuration" % (3 * n),
                               "stride_q_qd_u is the stride between configurations in d_q_qd_u",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "gravity is the gravity constant", "NUM_TIMESTEPS is the number of configurations"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(%d)" % (W * WAVE))
        self.gen_add_code_line("void forward_dynamics_gradient_kernel_coop8(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, "
                               "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "grid_tile_iter it(NUM_TIMESTEPS);              // lane / wave bookkeeping only: the tile loop below is per BLOCK",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % stage,
            "T *s_xch = reinterpret_cast<T *>(s_grid_dyn) + %d;" % xch_off,
            "const int nblocks = grid_num_blocks();",
            "const int bid = grid_block_id();",
            "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it; a wrong shape must not deadlock the barriers)" % (W * WAVE),
            "for (int k0 = bid*GRID_WAVE_SIZE; k0 < NUM_TIMESTEPS; k0 += nblocks*GRID_WAVE_SIZE){",
        ])
        self.indent_level += 1
        # every lane reads the few inputs its wave needs (its share of the input table, u of the bias-torque joints) straight from its
        # configuration's row: 8 waves staging all 3n inputs each was ~300 instructions per wave for ~12 values used (Atlas-30 K = 16384
        # 53.3 -> 48.5 us together with u - c published instead of c and u: profiles/r04/lean_store_path.txt).  Wave-uniform base of
        # the tile + one 32-bit per-lane row offset; lanes past the batch read the last valid row
        self.gen_add_code_line("const T *s_q_qd_u = grid_opaque_uniform(d_q_qd_u + (size_t)k0*stride_q_qd_u);")
        self.gen_add_code_line("const unsigned in_row = (unsigned)min(it.lane, NUM_TIMESTEPS - 1 - k0)*(unsigned)stride_q_qd_u*(unsigned)sizeof(T);")
        self.gen_add_code_line("const unsigned xb = (unsigned)reinterpret_cast<unsigned long long>(s_xch) + (unsigned)sizeof(T)*it.lane;")
        self.gen_add_code_line("const grid_in_lean<T> in = {s_q_qd_u, s_q_qd_u + %d, s_q_qd_u + %d, s_xch, it.lane, xb, xb + 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), "
                               "xb - 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), in_row};" % (n, 2 * n))
        self.gen_add_code_line("switch (it.wave_in_block){", True)
        for w, (cname, bases, pieces) in enumerate(names):
            self.gen_add_code_line("case %d: {" % w, True)
            if pieces:
                self.gen_add_code_line("grid_out_pieces<T,%d> out = {s_wave, d_df_du, k0, it.lane, it.W, NUM_TIMESTEPS};" % n_out)
            elif bases:
                self.gen_add_code_line("grid_out_runs<T,%d,%d,%s> out = {s_wave, d_df_du, k0, it.lane, it.W, NUM_TIMESTEPS};"
                                       % (n_out, n, ",".join(str(b) for b in bases)))
            else:
                self.gen_add_code_line("grid_out_ptr<T> out = {nullptr};     // (this core stores nothing)")
            self.gen_add_code_line("%s<T,C>(in, out, gravity);" % cname)
            self.gen_add_code_line("break;")
            self.gen_add_end_control_flow()
        self.gen_add_code_line("default: break;")
        self.gen_add_end_control_flow()
        self.gen_add_code_line("grid_block_sync();     // the exchange region and the staging regions are rewritten by the next tile")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the register-lean tile-cooperative forward-dynamics-gradient kernel (asynchronous, on `stream`)",
                              ["tile_blocks <= 0: one block per tile of 64 configurations (capped at 4*SUGGESTED_MAX_BLOCKS)",
                               "returns false when this robot has no such kernel"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_gradient_lean_launch(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, "
                               "const T gravity, const int num_timesteps, int tile_blocks, hipStream_t stream) {", True)
        self.gen_add_code_lines([
            "const size_t lds_bytes = (size_t)FD_DU_LEAN_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // > 64 KiB of dynamic LDS must be enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (lds_bytes > 65536 && configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_coop8<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));",
            "    configured_device = dev;",
            "}",
            "const int tiles = (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE;",
            "if (tile_blocks <= 0 || tile_blocks > tiles){tile_blocks = tiles;}",
            "if (tile_blocks > 4*SUGGESTED_MAX_BLOCKS){tile_blocks = 4*SUGGESTED_MAX_BLOCKS;}",
            "forward_dynamics_gradient_kernel_coop8<T><<<dim3(tile_blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(d_df_du,d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);" % (W * WAVE),
            "gpuErrchk(hipGetLastError());",
            "return true;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the register-lean tile-cooperative kernel", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_gradient_lean_attributes(hipFuncAttributes *attr) {", True)
        self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&forward_dynamics_gradient_kernel_coop8<T>))); return true;")
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # register-lean tile-cooperative FORWARD DYNAMICS (large robots): the prefix of the gradient kernel as a kernel of its own
    # ------------------------------------------------------------------------------------------
    def _lean_fd_prepare(self):
        """(slots, plan, stage, lds_elems) of `forward_dynamics_kernel_coop8`, or None (fp32, large robots, where the gradient's lean
        kernel exists: it is that kernel's prefix -- cores.lean_plan_fd)."""
        if hasattr(self, "_lean_fd_cache"):
            return self._lean_fd_cache
        self._lean_fd_cache = None
        if self.precision != "fp32" or self._lean_prepare() is None:
            return None
        W = cores.LEAN_WAVES
        slots, plan = cores.lean_plan_fd(self.spec, W, **self.lean_plan_options)
        stage = WAVE * 34
        self._lean_fd_cache = (slots, plan, stage, W * stage + WAVE * slots.count)
        return self._lean_fd_cache

    LEAN_FD_LAUNCH_SIG = ("bool forward_dynamics_lean_launch(T *d_qdd, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, "
                          "const T gravity, const int num_timesteps, int tile_blocks, hipStream_t stream)")

    def gen_forward_dynamics_lean_decl(self):
        """Forward declaration + constants of the register-lean forward-dynamics launcher, ahead of the reference-named host wrapper (a
        documented block of its own: no other kernel's object-cache key moves)."""
        prep = self._lean_fd_prepare()
        W = cores.LEAN_WAVES
        self.gen_add_func_doc("Launch the register-lean tile-cooperative forward-dynamics kernel (declaration; defined with the kernel at the end of the header)",
                              ["returns false when this robot / arithmetic has no such kernel (FD_LEAN_WAVES == 0)"], [], None)
        self.gen_add_code_lines(["template <typename T>", "__host__ inline", self.LEAN_FD_LAUNCH_SIG + ";"])
        if prep is None:
            self.gen_add_code_line("const int FD_LEAN_WAVES = 0; // no register-lean forward-dynamics kernel for this robot / arithmetic")
            self.gen_add_code_line("const int FD_LEAN_AUTO_MIN_TILES = 0;")
            self.gen_add_code_line("const int FD_LEAN_AUTO_MAX_TILES = 0;")
            self.gen_add_code_line("const int FD_LEAN_WAVE_MAX_K = 0;")
        else:
            slots, plan, stage, lds_elems = prep
            self.gen_add_code_line("const int FD_LEAN_WAVES = %d; // wavefronts per block of forward_dynamics_kernel_coop8 (block = %d threads, one tile)" % (W, W * WAVE))
            self.gen_add_code_line("const int FD_LEAN_AUTO_MIN_TILES = %d; // automatic choice of that kernel from this many tiles on (0: only on request)" % self.lean_fd_auto_min_tiles)
            # Atlas-30 (profiles/r04/lean_fd_sweep.txt): 12.4-13.0 us up to one tile per CU against 28 us for the lane-per-configuration
            # kernel (one wave per CU there); beyond two tiles per CU that kernel has a wave on every SIMD and wins (K = 65536: 37 against 49 us)
            self.gen_add_code_line("const int FD_LEAN_AUTO_MAX_TILES = %d; // ... and up to this many tiles (0: no upper limit)" % self.lean_fd_auto_max_tiles)
            self.gen_add_code_line("const int FD_LEAN_SHARED_MEM_COUNT = %d; // dynamic LDS in T elements: %d staging regions of %d + %d exchange slots x 64 lanes"
                                   % (lds_elems, W, stage, slots.count))
            self.gen_add_code_line("const int FD_LEAN_WAVE_MAX_K = %d; // the wave-per-configuration kernel automatically only up to this batch size (0: FD_WAVE_AUTO_MAX_K alone decides)" % self.lean_fd_wave_max_k)
        self.gen_add_code_line("")

    def gen_forward_dynamics_lean(self, use_thread_group=False):
        """`forward_dynamics_kernel_coop8`: qdd = Minv (u - c) of a large robot on the register-lean block of the gradient kernel -- its
        prefix as a kernel of its own (input table; Minv recursion once per base-rooted tree, shared through LDS, forward pass over all
        eight waves; bias torques depth first; qdd rows), then one wave writes the n accelerations.  The lane-per-configuration kernel
        runs the whole chain of a configuration on one lane: one wave per CU at K = 16384 for Atlas-30.  Reference mapping being
        replaced: algorithms/_forward_dynamics.py:21-112 (block per configuration)."""
        n, W = self.spec.n, cores.LEAN_WAVES
        prep = self._lean_fd_prepare()
        if prep is None:
            self.gen_add_func_doc("No register-lean forward-dynamics kernel for this robot / arithmetic", [], [], None)
            self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                     "bool forward_dynamics_lean_launch(T *, const T *, const int, const robotModel<T> *, const T, const int, int, hipStream_t) {return false;}",
                                     "template <typename T>", "__host__ inline",
                                     "bool forward_dynamics_lean_attributes(hipFuncAttributes *) {return false;}", ""])
            return
        slots, plan, stage, lds_elems = prep
        xch_off = W * stage
        names = []
        for w, (role, items) in enumerate(plan):
            cname = "forward_dynamics_lean_core_w%d" % w
            tr = cores.core_gradient_recompute(self.spec, "fd", cols=items, coop=(role, slots))
            self._emit_core(cname, "Register-lean tile-cooperative forward dynamics, wave %d of %d: %r%s" % (w, W, role, "; writes qdd" if role.out_qdd else ""),
                            tr, order="creation")
            names.append((cname, role.out_qdd))
        self.kernel_instances.append("__global__ void @NS::forward_dynamics_kernel_coop8<T>(T *, const T *, const int, "
                                     "const @NS::robotModel<T> *, const T, const int);")
        self.gen_add_func_doc("Computes forward dynamics (register-lean tile-cooperative: %d wavefronts, two per SIMD, share each tile of 64 configurations)" % W,
                              ["launch with EXACTLY %d threads per block and FD_LEAN_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS" % (W * WAVE),
                               "(use forward_dynamics_lean_launch); blocks grid-stride over the tiles"],
                              ["d_qdd is the output buffer, %d values per configuration" % n,
                               "d_q_qd_u is the input buffer, %d values read per configuration" % (3 * n),
                               "stride_q_qd_u is the stride between configurations in d_q_qd_u",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "gravity is the gravity constant", "NUM_TIMESTEPS is the number of configurations"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(%d)" % (W * WAVE))
        self.gen_add_code_line("void forward_dynamics_kernel_coop8(T *d_qdd, const T *d_q_qd_u, const int stride_q_qd_u, "
                               "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "grid_tile_iter it(NUM_TIMESTEPS);              // lane / wave bookkeeping only: the tile loop below is per BLOCK",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % stage,
            "T *s_xch = reinterpret_cast<T *>(s_grid_dyn) + %d;" % xch_off,
            "const int nblocks = grid_num_blocks();",
            "const int bid = grid_block_id();",
            "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it; a wrong shape must not deadlock the barriers)" % (W * WAVE),
            "for (int k0 = bid*GRID_WAVE_SIZE; k0 < NUM_TIMESTEPS; k0 += nblocks*GRID_WAVE_SIZE){",
        ])
        self.indent_level += 1
        self.gen_add_code_line("const T *s_q_qd_u = grid_opaque_uniform(d_q_qd_u + (size_t)k0*stride_q_qd_u);")
        self.gen_add_code_line("const unsigned in_row = (unsigned)min(it.lane, NUM_TIMESTEPS - 1 - k0)*(unsigned)stride_q_qd_u*(unsigned)sizeof(T);")
        self.gen_add_code_line("const unsigned xb = (unsigned)reinterpret_cast<unsigned long long>(s_xch) + (unsigned)sizeof(T)*it.lane;")
        self.gen_add_code_line("const grid_in_lean<T> in = {s_q_qd_u, s_q_qd_u + %d, s_q_qd_u + %d, s_xch, it.lane, xb, xb + 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), "
                               "xb - 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), in_row};" % (n, 2 * n))
        self.gen_add_code_line("switch (it.wave_in_block){", True)
        for w, (cname, writes) in enumerate(names):
            self.gen_add_code_line("case %d: {" % w, True)
            if writes:
                self.gen_add_code_line("grid_out_pieces<T,%d> out = {s_wave, d_qdd, k0, it.lane, it.W, NUM_TIMESTEPS};" % n)
            else:
                self.gen_add_code_line("grid_out_ptr<T> out = {nullptr};     // (this core stores nothing)")
            self.gen_add_code_line("%s<T,C>(in, out, gravity);" % cname)
            self.gen_add_code_line("break;")
            self.gen_add_end_control_flow()
        self.gen_add_code_line("default: break;")
        self.gen_add_end_control_flow()
        self.gen_add_code_line("grid_block_sync();     // the exchange region and the staging regions are rewritten by the next tile")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the register-lean tile-cooperative forward-dynamics kernel (asynchronous, on `stream`)",
                              ["tile_blocks <= 0: one block per tile of 64 configurations (capped at 4*SUGGESTED_MAX_BLOCKS)"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line(self.LEAN_FD_LAUNCH_SIG + " {", True)
        self.gen_add_code_lines([
            "const size_t lds_bytes = (size_t)FD_LEAN_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // > 64 KiB of dynamic LDS must be enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (lds_bytes > 65536 && configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&forward_dynamics_kernel_coop8<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));",
            "    configured_device = dev;",
            "}",
            "const int tiles = (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE;",
            "if (tile_blocks <= 0 || tile_blocks > tiles){tile_blocks = tiles;}",
            "if (tile_blocks > 4*SUGGESTED_MAX_BLOCKS){tile_blocks = 4*SUGGESTED_MAX_BLOCKS;}",
            "forward_dynamics_kernel_coop8<T><<<dim3(tile_blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(d_qdd,d_q_qd_u,stride_q_qd_u,d_robotModel,gravity,num_timesteps);" % (W * WAVE),
            "gpuErrchk(hipGetLastError());",
            "return true;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the register-lean forward-dynamics kernel", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool forward_dynamics_lean_attributes(hipFuncAttributes *attr) {", True)
        self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&forward_dynamics_kernel_coop8<T>))); return true;")
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # register-lean tile-cooperative DIRECT Minv (large robots): phases 0-2 of the gradient kernel, columns written from registers
    # ------------------------------------------------------------------------------------------
    def _lean_minv_prepare(self):
        if hasattr(self, "_lean_minv_cache"):
            return self._lean_minv_cache
        self._lean_minv_cache = None
        if self.precision != "fp32" or self._lean_prepare() is None:
            return None
        W = cores.LEAN_WAVES
        slots, plan = cores.lean_plan_minv(self.spec, W, **self.lean_plan_options)
        stage = WAVE * 34
        self._lean_minv_cache = (slots, plan, stage, W * stage + WAVE * slots.count)
        return self._lean_minv_cache

    LEAN_MINV_LAUNCH_SIG = ("bool direct_minv_lean_launch(T *d_Minv, const T *d_q, const int stride_q, const robotModel<T> *d_robotModel, "
                            "const int num_timesteps, int tile_blocks, hipStream_t stream)")

    def gen_direct_minv_lean_decl(self):
        """Forward declaration + constants of the register-lean direct-Minv launcher, ahead of the reference-named host wrapper."""
        prep = self._lean_minv_prepare()
        W = cores.LEAN_WAVES
        self.gen_add_func_doc("Launch the register-lean tile-cooperative direct-Minv kernel (declaration; defined with the kernel at the end of the header)",
                              ["returns false when this robot / arithmetic has no such kernel (MINV_LEAN_WAVES == 0)"], [], None)
        self.gen_add_code_lines(["template <typename T>", "__host__ inline", self.LEAN_MINV_LAUNCH_SIG + ";"])
        if prep is None:
            self.gen_add_code_lines(["const int MINV_LEAN_WAVES = 0; // no register-lean direct-Minv kernel for this robot / arithmetic",
                                     "const int MINV_LEAN_AUTO_MIN_TILES = 0;", "const int MINV_LEAN_AUTO_MAX_TILES = 0;", "const int MINV_LEAN_WAVE_MAX_K = 0;"])
        else:
            slots, plan, stage, lds_elems = prep
            self.gen_add_code_line("const int MINV_LEAN_WAVES = %d; // wavefronts per block of direct_minv_kernel_coop8 (block = %d threads, one tile)" % (W, W * WAVE))
            self.gen_add_code_line("const int MINV_LEAN_AUTO_MIN_TILES = %d; // automatic choice of that kernel from this many tiles on (0: only on request)" % self.lean_minv_auto[0])
            self.gen_add_code_line("const int MINV_LEAN_AUTO_MAX_TILES = %d; // ... and up to this many tiles (0: no upper limit)" % self.lean_minv_auto[1])
            self.gen_add_code_line("const int MINV_LEAN_SHARED_MEM_COUNT = %d; // dynamic LDS in T elements: %d staging regions of %d + %d exchange slots x 64 lanes"
                                   % (lds_elems, W, stage, slots.count))
            self.gen_add_code_line("const int MINV_LEAN_WAVE_MAX_K = %d; // the wave-per-configuration kernel automatically only up to this batch size (0: MINV_WAVE_AUTO_MAX_K alone decides)" % self.lean_minv_auto[2])
        self.gen_add_code_line("")

    def gen_direct_minv_lean(self, use_thread_group=False):
        """`direct_minv_kernel_coop8`: Minv (upper triangle) of a large robot on the register-lean block -- input table (sin q, cos q), the
        backward pass of the recursion once per base-rooted tree, the forward pass over all eight waves, every wave writing the columns it
        finishes from registers.  Reads q only (stride_q may be NUM_JOINTS).  Reference mapping being replaced: algorithms/_direct_minv.py:
        23-382 (block per configuration)."""
        n, W = self.spec.n, cores.LEAN_WAVES
        prep = self._lean_minv_prepare()
        if prep is None:
            self.gen_add_func_doc("No register-lean direct-Minv kernel for this robot / arithmetic", [], [], None)
            self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                     "bool direct_minv_lean_launch(T *, const T *, const int, const robotModel<T> *, const int, int, hipStream_t) {return false;}",
                                     "template <typename T>", "__host__ inline",
                                     "bool direct_minv_lean_attributes(hipFuncAttributes *) {return false;}", ""])
            return
        slots, plan, stage, lds_elems = prep
        xch_off = W * stage
        names = []
        for w, (role, items) in enumerate(plan):
            cname = "direct_minv_lean_core_w%d" % w
            tr = cores.core_gradient_recompute(self.spec, "fd", cols=items, coop=(role, slots))
            self._emit_core(cname, "Register-lean tile-cooperative direct Minv, wave %d of %d: %r; writes columns %s" % (w, W, role, role.minv_cols), tr, order="creation")
            names.append((cname, bool(role.minv_cols)))
        self.kernel_instances.append("__global__ void @NS::direct_minv_kernel_coop8<T>(T *, const T *, const int, const @NS::robotModel<T> *, const int);")
        self.gen_add_func_doc("Compute the inverse of the mass matrix (register-lean tile-cooperative: %d wavefronts, two per SIMD, share each tile of 64 configurations)" % W,
                              ["launch with EXACTLY %d threads per block and MINV_LEAN_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS" % (W * WAVE),
                               "(use direct_minv_lean_launch); blocks grid-stride over the tiles"],
                              ["d_Minv is the output buffer, %d values per configuration (column-major, upper triangle)" % (n * n),
                               "d_q is the input buffer, %d values read per configuration" % n,
                               "stride_q is the stride between configurations in d_q",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "NUM_TIMESTEPS is the number of configurations"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(%d)" % (W * WAVE))
        self.gen_add_code_line("void direct_minv_kernel_coop8(T *d_Minv, const T *d_q, const int stride_q, const robotModel<T> *d_robotModel, const int NUM_TIMESTEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "grid_tile_iter it(NUM_TIMESTEPS);              // lane / wave bookkeeping only: the tile loop below is per BLOCK",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % stage,
            "T *s_xch = reinterpret_cast<T *>(s_grid_dyn) + %d;" % xch_off,
            "const int nblocks = grid_num_blocks();",
            "const int bid = grid_block_id();",
            "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it; a wrong shape must not deadlock the barriers)" % (W * WAVE),
            "for (int k0 = bid*GRID_WAVE_SIZE; k0 < NUM_TIMESTEPS; k0 += nblocks*GRID_WAVE_SIZE){",
        ])
        self.indent_level += 1
        self.gen_add_code_line("const T *s_q = grid_opaque_uniform(d_q + (size_t)k0*stride_q);       // (q only: the cores of this kernel read nothing else)")
        self.gen_add_code_line("const unsigned in_row = (unsigned)min(it.lane, NUM_TIMESTEPS - 1 - k0)*(unsigned)stride_q*(unsigned)sizeof(T);")
        self.gen_add_code_line("const unsigned xb = (unsigned)reinterpret_cast<unsigned long long>(s_xch) + (unsigned)sizeof(T)*it.lane;")
        self.gen_add_code_line("const grid_in_lean<T> in = {s_q, nullptr, nullptr, s_xch, it.lane, xb, xb + 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), "
                               "xb - 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), in_row};")
        self.gen_add_code_line("switch (it.wave_in_block){", True)
        for w, (cname, writes) in enumerate(names):
            self.gen_add_code_line("case %d: {" % w, True)
            if writes:
                self.gen_add_code_line("grid_out_pieces<T,%d> out = {s_wave, d_Minv, k0, it.lane, it.W, NUM_TIMESTEPS};" % (n * n))
            else:
                self.gen_add_code_line("grid_out_ptr<T> out = {nullptr};     // (this core stores nothing)")
            self.gen_add_code_line("%s<T,C>(in, out, static_cast<T>(0));" % cname)
            self.gen_add_code_line("break;")
            self.gen_add_end_control_flow()
        self.gen_add_code_line("default: break;")
        self.gen_add_end_control_flow()
        self.gen_add_code_line("grid_block_sync();     // the exchange region and the staging regions are rewritten by the next tile")
        self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the register-lean tile-cooperative direct-Minv kernel (asynchronous, on `stream`)",
                              ["tile_blocks <= 0: one block per tile of 64 configurations (capped at 4*SUGGESTED_MAX_BLOCKS)"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line(self.LEAN_MINV_LAUNCH_SIG + " {", True)
        self.gen_add_code_lines([
            "const size_t lds_bytes = (size_t)MINV_LEAN_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // > 64 KiB of dynamic LDS must be enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (lds_bytes > 65536 && configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&direct_minv_kernel_coop8<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));",
            "    configured_device = dev;",
            "}",
            "const int tiles = (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE;",
            "if (tile_blocks <= 0 || tile_blocks > tiles){tile_blocks = tiles;}",
            "if (tile_blocks > 4*SUGGESTED_MAX_BLOCKS){tile_blocks = 4*SUGGESTED_MAX_BLOCKS;}",
            "direct_minv_kernel_coop8<T><<<dim3(tile_blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(d_Minv,d_q,stride_q,d_robotModel,num_timesteps);" % (W * WAVE),
            "gpuErrchk(hipGetLastError());",
            "return true;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the register-lean direct-Minv kernel", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool direct_minv_lean_attributes(hipFuncAttributes *attr) {", True)
        self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&direct_minv_kernel_coop8<T>))); return true;")
        self.gen_add_end_function()

    # ------------------------------------------------------------------------------------------
    # register-lean tile-cooperative INVERSE-dynamics gradient (large robots, qdd = 0 variant)
    # ------------------------------------------------------------------------------------------
    def _lean_id_prepare(self):
        """(slots, plan, stage, lds_elems) of `inverse_dynamics_gradient_kernel_coop8`, or None (same conditions as the forward-dynamics
        kernel's: large robot, fp32, recomputing schedule)."""
        if hasattr(self, "_lean_id_cache"):
            return self._lean_id_cache
        n, W = self.spec.n, cores.LEAN_WAVES
        self._lean_id_cache = None
        if n < self.lean_min_joints or self.precision not in ("fp32", "mixed") or (self.grad_schedule != "recompute" and n > 12):
            return None             # (mixed: this kernel has no double part -- the same arithmetic as in the fp32 library)
        slots, plan = cores.lean_plan_id(self.spec, False, W, **self.lean_id_plan_options)
        if isinstance(self.lean_probe, str) and self.lean_probe.startswith("id_only:"):       # (experiment: one wave keeps its columns)
            keep = int(self.lean_probe.split(":")[1])
            plan = [(role, items if w == keep else []) for w, (role, items) in enumerate(plan)]
        stage = WAVE * 34
        lds_elems = W * stage + WAVE * slots.count
        if 4 * lds_elems > 160 * 1024:
            self.note("no register-lean 8-wave inverse-dynamics-gradient kernel (ID_DU_LEAN_WAVES = 0): input table + 8 staging regions need "
                      "%d KB of the CU's 160 KB of LDS" % (4 * lds_elems // 1024))
            return None
        self._lean_id_cache = (slots, plan, stage, lds_elems)
        return self._lean_id_cache

    LEAN_ID_LAUNCH_SIG = ("bool inverse_dynamics_gradient_lean_launch(T *d_dc_du, const T *d_q_qd, const int stride_q_qd, const robotModel<T> *d_robotModel, "
                          "const T gravity, const int num_timesteps, int tile_blocks, hipStream_t stream)")

    def gen_inverse_dynamics_gradient_lean_decl(self):
        """Forward declaration of the register-lean inverse-dynamics-gradient launcher and its constants, ahead of the reference-named host
        wrapper that dispatches it; the kernel itself is emitted last (gen_inverse_dynamics_gradient_lean).  A documented block of its
        own: the object-cache keys of the other kernels do not move (host.kernel_dependency_hashes)."""
        prep = self._lean_id_prepare()
        W = cores.LEAN_WAVES
        self.gen_add_func_doc("Launch the register-lean tile-cooperative inverse-dynamics-gradient kernel, qdd = 0 (declaration; defined with the kernel at the end of the header)",
                              ["returns false when this robot / arithmetic has no such kernel (ID_DU_LEAN_WAVES == 0)"], [], None)
        self.gen_add_code_lines(["template <typename T>", "__host__ inline", self.LEAN_ID_LAUNCH_SIG + ";"])
        if prep is None:
            self.gen_add_code_line("const int ID_DU_LEAN_WAVES = 0; // no register-lean inverse-dynamics-gradient kernel for this robot / arithmetic")
            self.gen_add_code_line("const int ID_DU_LEAN_AUTO_MIN_TILES = 0;")
            self.gen_add_code_line("const int ID_DU_LEAN_WAVE_MAX_K = 0;")
        else:
            slots, plan, stage, lds_elems = prep
            self.gen_add_code_line("const int ID_DU_LEAN_WAVES = %d; // wavefronts per block of inverse_dynamics_gradient_kernel_coop8 (block = %d threads, one tile)" % (W, W * WAVE))
            self.gen_add_code_line("const int ID_DU_LEAN_AUTO_MIN_TILES = %d; // automatic choice of that kernel from this many tiles on (0: only on request)" % self.lean_id_auto_min_tiles)
            self.gen_add_code_line("const int ID_DU_LEAN_SHARED_MEM_COUNT = %d; // dynamic LDS in T elements: %d staging regions of %d + %d table slots x 64 lanes"
                                   % (lds_elems, W, stage, slots.count))
            # Atlas-30: wave-per-configuration 15.9 us at K = 1024, 34.5 at 2048 against 23.8 / 24.7 us here (same file)
            self.gen_add_code_line("const int ID_DU_LEAN_WAVE_MAX_K = %d; // the wave-per-configuration kernel automatically only up to this batch size (0: ID_DU_WAVE_AUTO_MAX_K alone decides)" % self.lean_id_wave_max_k)
        self.gen_add_code_line("")

    def gen_inverse_dynamics_gradient_lean(self, use_thread_group=False):
        """`inverse_dynamics_gradient_kernel_coop8` (qdd = 0): the inverse-dynamics gradient of a large robot on the register-lean block
        of the forward-dynamics kernel -- eight wavefronts per tile of 64 configurations, two per SIMD, at most 256 registers each; a
        block-shared input table (sin q, cos q, qd), ONE barrier, then every wave's contiguous runs of gradient half-columns, cut at the
        32-byte sectors of the output row (cores.lean_plan_id, cores.AlignedPieces).  Replaces the 4-way column split for batches that
        fill the chip (each of its waves owned a SIMD, staged all inputs and spilled).  Reference mapping being replaced:
        algorithms/_inverse_dynamics_gradient.py:199-246,501-540 (threads of a block = gradient columns)."""
        n, W = self.spec.n, cores.LEAN_WAVES
        prep = self._lean_id_prepare()
        if prep is None:
            self.gen_add_func_doc("No register-lean inverse-dynamics-gradient kernel for this robot / arithmetic", [], [], None)
            self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                     "bool inverse_dynamics_gradient_lean_launch(T *, const T *, const int, const robotModel<T> *, const T, const int, int, hipStream_t) {return false;}",
                                     "template <typename T>", "__host__ inline",
                                     "bool inverse_dynamics_gradient_lean_attributes(hipFuncAttributes *) {return false;}", ""])
            return
        slots, plan, stage, lds_elems = prep
        xch_off = W * stage
        self.lean_id_stats = dict(plan=[(repr(r), list(items)) for (r, items) in plan], slots=slots.count, lds_bytes=4 * lds_elems, model=dict(slots.lean_model))
        names = []
        for w, (role, items) in enumerate(plan):
            cname = "inverse_dynamics_gradient_lean_core_w%d" % w
            tr = cores.core_gradient_recompute(self.spec, "id", cols=items, coop=(role, slots))
            self._emit_core(cname, "Register-lean tile-cooperative inverse-dynamics gradient (qdd = 0), wave %d of %d: input table of joints %s; gradient "
                            "half-columns (column, 0 = d/dq | 1 = d/dqd) %s" % (w, W, role.joints, list(items)), tr, order="creation")
            names.append(cname)
        n_out = self.io_layout["ID_DU"]["n_out"]
        self.kernel_instances.append("__global__ void @NS::inverse_dynamics_gradient_kernel_coop8<T>(T *, const T *, const int, "
                                     "const @NS::robotModel<T> *, const T, const int);")
        self.gen_add_func_doc("Computes the gradient of inverse dynamics, qdd = 0 (register-lean tile-cooperative: %d wavefronts, two per SIMD, share each tile of 64 configurations)" % W,
                              ["launch with EXACTLY %d threads per block and ID_DU_LEAN_SHARED_MEM_COUNT*sizeof(T) of dynamic LDS" % (W * WAVE),
                               "(use inverse_dynamics_gradient_lean_launch); blocks grid-stride over the tiles",
                               "LDS: [%d staging regions | input table]" % W],
                              ["d_dc_du is the output buffer, %d values per configuration" % n_out,
                               "d_q_qd is the input buffer, %d values read per configuration" % (2 * n),
                               "stride_q_qd is the stride between configurations in d_q_qd",
                               "d_robotModel is the pointer to the initialized model specific helpers on the GPU (unused: constants are baked in)",
                               "gravity is the gravity constant", "NUM_TIMESTEPS is the number of configurations"], None)
        self.gen_add_code_line("template <typename T, typename C = typename grid_compute<T>::type>")
        self.gen_add_code_line("__global__ __launch_bounds__(%d)" % (W * WAVE))
        self.gen_add_code_line("void inverse_dynamics_gradient_kernel_coop8(T *d_dc_du, const T *d_q_qd, const int stride_q_qd, "
                               "const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS) {", True)
        self.gen_add_code_lines([
            "extern __shared__ __align__(16) unsigned char s_grid_dyn[];",
            "grid_tile_iter it(NUM_TIMESTEPS);              // lane / wave bookkeeping only: the tile loop below is per BLOCK",
            "T *s_wave = reinterpret_cast<T *>(s_grid_dyn) + it.wave_in_block*%d;" % stage,
            "T *s_xch = reinterpret_cast<T *>(s_grid_dyn) + %d;" % xch_off,
            "const int nblocks = grid_num_blocks();",
            "const int bid = grid_block_id();",
            "if (grid_block_threads() != %d){return;}    // (the launcher guarantees it; a wrong shape must not deadlock the barrier)" % (W * WAVE),
        ])
        setup = ["// every lane reads the few inputs its wave needs from its row: wave-uniform base of the tile + one 32-bit per-lane row offset",
                 "const T *s_q_qd = grid_opaque_uniform(d_q_qd + (size_t)k0*stride_q_qd);",
                 "const unsigned in_row = (unsigned)min(it.lane, NUM_TIMESTEPS - 1 - k0)*(unsigned)stride_q_qd*(unsigned)sizeof(T);",
                 "const unsigned xb = (unsigned)reinterpret_cast<unsigned long long>(s_xch) + (unsigned)sizeof(T)*it.lane;",
                 "const grid_in_lean<T> in = {s_q_qd, s_q_qd + %d, nullptr, s_xch, it.lane, xb, xb + 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), "
                 "xb - 256u*GRID_WAVE_SIZE*(unsigned)sizeof(T), in_row};" % n]
        loop = "for (int k0 = bid*GRID_WAVE_SIZE; k0 < NUM_TIMESTEPS; k0 += nblocks*GRID_WAVE_SIZE){"
        sync = "grid_block_sync();     // the input table and the staging regions are rewritten by the next tile"
        if self.lean_loop_per_role:
            # one tile loop PER ROLE (the switch outside): what hipcc hoists out of a loop then belongs to one role, not to all eight
            self.gen_add_code_line("switch (it.wave_in_block){", True)
            for w, cname in enumerate(names):
                self.gen_add_code_line("case %d: {" % w, True)
                self.gen_add_code_line(loop, True)
                self.gen_add_code_lines(setup)
                self.gen_add_code_line("grid_out_pieces<T,%d> out = {s_wave, d_dc_du, k0, it.lane, it.W, NUM_TIMESTEPS};" % n_out)
                self.gen_add_code_line("%s<T,C>(in, out, gravity);" % cname)
                self.gen_add_code_line(sync)
                self.gen_add_end_control_flow()
                self.gen_add_code_line("break;")
                self.gen_add_end_control_flow()
            self.gen_add_code_line("default: break;")
            self.gen_add_end_control_flow()
        else:
            self.gen_add_code_line(loop)
            self.indent_level += 1
            self.gen_add_code_lines(setup)
            self.gen_add_code_line("switch (it.wave_in_block){", True)
            for w, cname in enumerate(names):
                self.gen_add_code_line("case %d: {" % w, True)
                self.gen_add_code_line("grid_out_pieces<T,%d> out = {s_wave, d_dc_du, k0, it.lane, it.W, NUM_TIMESTEPS};" % n_out)
                self.gen_add_code_line("%s<T,C>(in, out, gravity);" % cname)
                self.gen_add_code_line("break;")
                self.gen_add_end_control_flow()
            self.gen_add_code_line("default: break;")
            self.gen_add_end_control_flow()
            self.gen_add_code_line(sync)
            self.gen_add_end_control_flow()
        self.gen_add_end_function()
        self.gen_add_func_doc("Launch the register-lean tile-cooperative inverse-dynamics-gradient kernel (asynchronous, on `stream`)",
                              ["tile_blocks <= 0: one block per tile of 64 configurations (capped at 4*SUGGESTED_MAX_BLOCKS)"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line(self.LEAN_ID_LAUNCH_SIG + " {", True)
        self.gen_add_code_lines([
            "const size_t lds_bytes = (size_t)ID_DU_LEAN_SHARED_MEM_COUNT*sizeof(T);",
            "static thread_local int configured_device = -1;        // > 64 KiB of dynamic LDS must be enabled once per device",
            "int dev = 0; gpuErrchk(hipGetDevice(&dev));",
            "if (lds_bytes > 65536 && configured_device != dev){",
            "    gpuErrchk(hipFuncSetAttribute(reinterpret_cast<const void *>(&inverse_dynamics_gradient_kernel_coop8<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));",
            "    configured_device = dev;",
            "}",
            "const int tiles = (num_timesteps + GRID_WAVE_SIZE - 1)/GRID_WAVE_SIZE;",
            "if (tile_blocks <= 0 || tile_blocks > tiles){tile_blocks = tiles;}",
            "if (tile_blocks > 4*SUGGESTED_MAX_BLOCKS){tile_blocks = 4*SUGGESTED_MAX_BLOCKS;}",
            "inverse_dynamics_gradient_kernel_coop8<T><<<dim3(tile_blocks,1,1),dim3(%d,1,1),lds_bytes,stream>>>(d_dc_du,d_q_qd,stride_q_qd,d_robotModel,gravity,num_timesteps);" % (W * WAVE),
            "gpuErrchk(hipGetLastError());",
            "return true;",
        ])
        self.gen_add_end_function()
        self.gen_add_func_doc("hipFuncGetAttributes of the register-lean inverse-dynamics-gradient kernel", ["returns false when this robot has none"], [], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__ inline")
        self.gen_add_code_line("bool inverse_dynamics_gradient_lean_attributes(hipFuncAttributes *attr) {", True)
        self.gen_add_code_line("gpuErrchk(hipFuncGetAttributes(attr, reinterpret_cast<const void *>(&inverse_dynamics_gradient_kernel_coop8<T>))); return true;")
        self.gen_add_end_function()

    def _emit_no_coop(self):
        self.gen_add_code_line("const int FD_DU_COOP_SHARED_MEM_COUNT = 0;")
        self.gen_add_code_line("const int FD_DU_COOP_AUTO_MIN_TILES = 0;")
        self.gen_add_code_lines(["template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_coop_launch(T *, const T *, const int, const robotModel<T> *, const T, const int, int, hipStream_t) {return false;}",
                                 "template <typename T>", "__host__ inline",
                                 "bool forward_dynamics_gradient_coop_attributes(hipFuncAttributes *) {return false;}", ""])
