"""Emission of the spatial-algebra device library: ``dot_prod``, ``mx0..mx5`` (+ ``_peq``, ``_scaled``, ``_peq_scaled``), the
runtime-selected ``mxX*`` family, ``fx``, ``fx_zeroed``, ``fx_times_v`` and ``fx_times_v_peq``.

Public surface of the reference's header (helpers/_spatial_algebra_helpers.py:35-257: same names, argument orders and the
column-major 6x6 layout ``s_matX[6*col + row]``), so that user kernels written against it compile against this header too.
The generated kernels themselves never call these functions -- their 6x6 algebra is folded entry by entry into straight-line
code by the tracer (emit/algorithms.py: mxS, fxv) -- which is why they are thin ``__host__ __device__`` templates here.

Nothing below is a table copied from the reference: every statement is derived from the definition of the spatial cross
products.  With a spatial vector ``x = [w; v]`` (angular part first):

    crm(x) = [[ skew(w), 0 ], [ skew(v), skew(w) ]]        motion cross product,   crm(x) y = x x y
    crf(x) = -crm(x)^T                                     force cross product,    crf(x) f = x x* f
    skew(a) b = a x b,  skew(a)[i][j] = -eps_ijk a_k

``mxK(dst, x)`` is column K of crm(x), i.e. x x e_K; ``fx(dst, x)`` is the whole matrix crf(x).
"""


def _skew(base):
    """3x3 cross-product matrix of the 3-vector held at indices base..base+2: entries (sign, index) or None."""
    m = [[None] * 3 for _ in range(3)]
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        m[i][j] = (-1, base + k)        # (a x b)_i = a_j b_k - a_k b_j
        m[i][k] = (+1, base + j)
    return m


def crm_entries():
    """crm(x)[row][col] as (sign, index into x) or None (structural zero)."""
    w, v = _skew(0), _skew(3)
    out = [[None] * 6 for _ in range(6)]
    for r in range(3):
        for c in range(3):
            out[r][c] = w[r][c]
            out[3 + r][c] = v[r][c]
            out[3 + r][3 + c] = w[r][c]
    return out


def crf_entries():
    """crf(x) = -crm(x)^T."""
    m = crm_entries()
    return [[(None if m[c][r] is None else (-m[c][r][0], m[c][r][1])) for c in range(6)] for r in range(6)]


def _term(entry, src, first):
    sign, idx = entry
    if first:
        return ("-" if sign < 0 else "") + "%s[%d]" % (src, idx)
    return (" - " if sign < 0 else " + ") + "%s[%d]" % (src, idx)


class SpatialAlgebraEmitMixin:
    def gen_spatial_algebra_helpers(self):
        # dot_prod: the four const / non-const pointer combinations of the reference's overload set
        for (q1, q2) in (("const ", "const "), ("", "const "), ("const ", ""), ("", "")):
            self.gen_add_func_doc("Compute the dot product between two vectors", ["evaluated by the calling lane"],
                                  ["vec1 is the first vector of length N with stride S1", "vec2 is the second vector of length N with stride S2"],
                                  "the resulting final value")
            self.gen_add_code_lines(["template <typename T, int N, int S1, int S2>", "__host__ __device__ __forceinline__",
                                     "T dot_prod(%sT *vec1, %sT *vec2) {" % (q1, q2)], True)
            self.gen_add_code_lines(["T result = 0;", "#pragma unroll", "for (int i = 0; i < N; i++){result += vec1[i*S1] * vec2[i*S2];}", "return result;"])
            self.gen_add_end_function()
        crm = crm_entries()
        for k in range(6):
            col = [crm[r][k] for r in range(6)]
            for (suffix, accumulate, scaled) in (("", False, False), ("_peq", True, False), ("_scaled", False, True), ("_peq_scaled", True, True)):
                what = ("Adds" if accumulate else "Generates") + " the motion vector cross product matrix column %d" % k
                params = ["s_vecX is the destination vector", "s_vec is the source vector"] + (["alpha is the scaling factor"] if scaled else [])
                self.gen_add_func_doc(what, ["column %d of crm(s_vec), i.e. s_vec x e_%d%s" % (k, k, "; structural zeros are left untouched" if accumulate else "")],
                                      params, None)
                self.gen_add_code_lines(["template <typename T>", "__host__ __device__ __forceinline__",
                                         "void mx%d%s(T *s_vecX, const T *s_vec%s) {" % (k, suffix, ", const T alpha" if scaled else "")], True)
                for r in range(6):
                    if col[r] is None:
                        if not accumulate:
                            self.gen_add_code_line("s_vecX[%d] = static_cast<T>(0);" % r)
                        continue
                    rhs = _term(col[r], "s_vec", True) + ("*alpha" if scaled else "")
                    self.gen_add_code_line("s_vecX[%d] %s %s;" % (r, "+=" if accumulate else "=", rhs))
                self.gen_add_end_function()
        for (suffix, scaled) in (("", False), ("_peq", False), ("_scaled", True), ("_peq_scaled", True)):
            params = ["s_vecX is the destination vector", "s_vec is the source vector"] + (["alpha is the scaling factor"] if scaled else []) \
                + ["S_ind selects the column (0-2 revolute x/y/z, 3-5 prismatic x/y/z)"]
            self.gen_add_func_doc("Generates the motion vector cross product matrix for a runtime selected column",
                                  ["a wave-uniform S_ind keeps the switch a scalar branch"], params, None)
            self.gen_add_code_lines(["template <typename T>", "__host__ __device__ __forceinline__",
                                     "void mxX%s(T *s_vecX, const T *s_vec, %sconst int S_ind) {" % (suffix, "const T alpha, " if scaled else "")], True)
            self.gen_add_code_line("switch(S_ind){", True)
            for k in range(6):
                self.gen_add_code_line("case %d: mx%d%s<T>(s_vecX, s_vec%s); break;" % (k, k, suffix, ", alpha" if scaled else ""))
            self.gen_add_code_line("default: break;")
            self.gen_add_end_control_flow()
            self.gen_add_end_function()
        crf = crf_entries()
        for (name, zeroed) in (("fx", False), ("fx_zeroed", True)):
            self.gen_add_func_doc("Generates the force vector cross product matrix" + (" for a pre-zeroed destination" if zeroed else ""),
                                  ["s_matX = crf(s_vecX) = -crm(s_vecX)^T, column-major 6x6 (entry [row][col] at 6*col + row)"]
                                  + (["Assumes destination is zeroed"] if zeroed else []),
                                  ["s_matX is the destination matrix", "s_vecX is the source vector"], None)
            self.gen_add_code_lines(["template <typename T>", "__host__ __device__ __forceinline__", "void %s(T *s_matX, const T *s_vecX) {" % name], True)
            for c in range(6):
                for r in range(6):
                    if crf[r][c] is None:
                        if not zeroed:
                            self.gen_add_code_line("s_matX[6*%d + %d] = static_cast<T>(0);" % (c, r))
                    else:
                        self.gen_add_code_line("s_matX[6*%d + %d] = %s;" % (c, r, _term(crf[r][c], "s_vecX", True)))
            self.gen_add_end_function()
        for (name, accumulate) in (("fx_times_v", False), ("fx_times_v_peq", True)):
            self.gen_add_func_doc(("Adds" if accumulate else "Generates") + " the force vector cross product matrix multiplied by the input vector",
                                  ["s_result %s crf(s_fxVec) s_timesVec" % ("+=" if accumulate else "=")],
                                  ["s_result is the result vector", "s_fxVec is the fx vector", "s_timesVec is the multipled vector"], None)
            self.gen_add_code_lines(["template <typename T>", "__host__ __device__ __forceinline__",
                                     "void %s(T *s_result, const T *s_fxVec, const T *s_timesVec) {" % name], True)
            for r in range(6):
                terms = [(crf[r][c][0], crf[r][c][1], c) for c in range(6) if crf[r][c] is not None]
                rhs = ""
                for i, (sign, idx, c) in enumerate(terms):
                    rhs += (("-" if sign < 0 else "") if i == 0 else (" - " if sign < 0 else " + ")) + "s_fxVec[%d]*s_timesVec[%d]" % (idx, c)
                self.gen_add_code_line("s_result[%d] %s %s;" % (r, "+=" if accumulate else "=", rhs))
            self.gen_add_end_function()
