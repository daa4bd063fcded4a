"""CPU evaluation of the generated algorithms, under the reference's method names.

The reference generator carries a numpy implementation of its algorithms as methods
(``test_rnea``, ``test_minv``, ``test_rnea_grad``, ``test_fd_grad``, ... -- reference ``_test.py:109,213,490,496``) that
GRiD-style test programs call to check a header.  Here the same entry points are served by the generator's own
intermediate representation: the traced core that is emitted as HIP text is *interpreted* with numpy in float64
(``emit/trace.py: Tracer.evaluate``).  So these methods check exactly the arithmetic the kernels execute (same sparsity
specialisation, same operation order), without a compiler or a GPU.  They are not the test oracle of this repository
(``oracle/`` is, and it shares no code with this module).

Conventions follow the reference: one configuration per call (1-D ``q, qd, qdd, u``), ``GRAVITY = -9.81`` means gravity
pulls along -z (the kernels' ``gravity`` argument is ``-GRAVITY``), ``v, a, f`` are ``6 x n`` with ``f`` accumulated over
subtrees, ``Minv[r, c]``, gradients are ``n x 2n = [d/dq | d/dqd]``.  For prismatic joints the gradients use the force
cross product (the reference's ``_test.py:311,437`` applies the motion cross product there).
"""
import numpy as np

from .emit import cores


class VerificationMixin:
    def _trace_cached(self, key, builder):
        cache = self.__dict__.setdefault("_verification_traces", {})
        if key not in cache:
            cache[key] = builder()
        return cache[key]

    def _evaluate(self, tr, q, qd=None, qdd=None, u=None, gravity=0.0):
        n = self.spec.n
        inputs = {"gravity": np.array([float(gravity)])}
        for name, vec in (("q", q), ("qd", qd), ("qdd", qdd), ("u", u)):
            if vec is not None:
                vec = np.asarray(vec, dtype=np.float64).reshape(n)
                for j in range(n):
                    inputs["in.%s(%d)" % (name, j)] = vec[j:j + 1]
        outs = tr.evaluate(inputs, "float64")
        size = 1 + max(int(dst) for (dst, _) in tr.outputs if not isinstance(dst, str))
        flat = np.zeros(size)
        for (dst, _), val in zip(tr.outputs, outs):
            if not isinstance(dst, str):
                flat[int(dst)] = float(np.asarray(val).reshape(-1)[0])
        return flat

    # ---- reference _test.py:109 ----
    def test_rnea(self, q, qd, qdd=None, GRAVITY=-9.81):
        """(c, v, a, f): bias/inverse-dynamics torques and the per-joint spatial velocity, acceleration, force (6 x n)."""
        n = self.spec.n
        use_qdd = qdd is not None
        c = self._evaluate(self._trace_cached(("id", use_qdd), lambda: cores.core_inverse_dynamics(self.spec, use_qdd)),
                           q, qd, qdd, gravity=-GRAVITY)
        vaf = self._evaluate(self._trace_cached(("vaf", use_qdd), lambda: cores.core_inverse_dynamics_vaf(self.spec, use_qdd)),
                             q, qd, qdd, gravity=-GRAVITY)
        v, a, f = (vaf[k * 6 * n:(k + 1) * 6 * n].reshape(n, 6).T.copy() for k in range(3))
        return (c, v, a, f)

    # ---- reference _test.py:204,213 ----
    def test_densify_Minv(self, Minv):
        Minv = np.array(Minv, dtype=np.float64)
        return np.triu(Minv) + np.triu(Minv, 1).T

    def test_minv(self, q, output_dense=True):
        """M^-1(q): upper triangle (zeros below the diagonal) or, by default, the dense symmetric matrix."""
        n = self.spec.n
        flat = self._evaluate(self._trace_cached(("minv",), lambda: cores.core_direct_minv(self.spec)), q)
        Minv = flat.reshape(n, n).T.copy()         # kernel layout is column-major: flat[n*c + r] = Minv[r, c]
        return self.test_densify_Minv(Minv) if output_dense else Minv

    # ---- reference _test.py:490 ----
    def test_rnea_grad(self, q, qd, qdd=None, GRAVITY=-9.81):
        """dc_du = [dc/dq | dc/dqd] (n x 2n) at (q, qd, qdd)."""
        n = self.spec.n
        use_qdd = qdd is not None
        flat = self._evaluate(self._trace_cached(("idg", use_qdd), lambda: cores.core_inverse_dynamics_gradient(self.spec, use_qdd)),
                              q, qd, qdd, gravity=-GRAVITY)
        return np.hstack([flat[:n * n].reshape(n, n).T, flat[n * n:].reshape(n, n).T])

    # ---- reference _test.py:496 ----
    def test_fd_grad(self, q, qd, u, GRAVITY=-9.81):
        """df_du = [dqdd/dq | dqdd/dqd] (n x 2n) = -Minv dc_du at qdd = Minv (u - c)."""
        n = self.spec.n
        flat = self._evaluate(self._trace_cached(("fdg",), lambda: cores.core_forward_dynamics_gradient(self.spec, False)),
                              q, qd, u=u, gravity=-GRAVITY)
        return np.hstack([flat[:n * n].reshape(n, n).T, flat[n * n:].reshape(n, n).T])

    def test_forward_dynamics(self, q, qd, u, GRAVITY=-9.81):
        """qdd = Minv (u - c)  (no reference counterpart; what forward_dynamics_kernel computes)."""
        return self._evaluate(self._trace_cached(("fd",), lambda: cores.core_forward_dynamics(self.spec)), q, qd, u=u, gravity=-GRAVITY)

    # ---- spatial cross-product helpers (reference _test.py:522-681), 6-vectors [angular; linear] ----
    @staticmethod
    def fx(vec):
        """6x6 force cross-product matrix crf(v) = [[w~, v~], [0, w~]]."""
        w, v = np.asarray(vec, dtype=np.float64).reshape(6)[:3], np.asarray(vec, dtype=np.float64).reshape(6)[3:]
        sk = lambda x: np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])
        out = np.zeros((6, 6))
        out[:3, :3] = sk(w); out[:3, 3:] = sk(v); out[3:, 3:] = sk(w)
        return out

    def mx(self, vec):
        """6x6 motion cross-product matrix crm(v) = -crf(v)^T."""
        return -self.fx(vec).transpose()

    def fxv(self, fxVec, timesVec):
        return self.fx(fxVec) @ np.asarray(timesVec, dtype=np.float64).reshape(6)

    def mxv(self, fxVec, timesVec):
        return self.mx(fxVec) @ np.asarray(timesVec, dtype=np.float64).reshape(6)

    def mxS(self, S, vec, alpha=1.0):
        """crm(vec) S alpha for a motion subspace vector S."""
        return alpha * (self.mx(vec) @ np.asarray(S, dtype=np.float64).reshape(6))

    def fxS(self, S, vec, alpha=1.0):
        return -self.mxS(S, vec, alpha)

    def _mx_unit(self, k, vec, alpha):
        e = np.zeros(6); e[k] = 1.0
        return self.mxS(e, vec, alpha)

    def mx0(self, vec, alpha=1.0): return self._mx_unit(0, vec, alpha)
    def mx1(self, vec, alpha=1.0): return self._mx_unit(1, vec, alpha)
    def mx2(self, vec, alpha=1.0): return self._mx_unit(2, vec, alpha)
    def mx3(self, vec, alpha=1.0): return self._mx_unit(3, vec, alpha)
    def mx4(self, vec, alpha=1.0): return self._mx_unit(4, vec, alpha)
    def mx5(self, vec, alpha=1.0): return self._mx_unit(5, vec, alpha)
