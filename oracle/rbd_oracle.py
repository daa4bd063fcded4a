"""CPU oracle: numpy float64 restatement of the reference's rigid-body-dynamics algorithms.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``gridcodegenerator_amd/``) imports this
module; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` as the checker / reported CPU baseline, never as the thing shipped.

What it restates: the numpy reference algorithms the generator authors mixed into the
``GRiDCodeGenerator`` class (reference ``_test.py``), which are the in-container statement of
``rbdReference``:

    rnea_fpass / rnea_bpass / rnea      <- _test.py:5-76, 78-107, 109-115
    minv_bpass / minv_fpass / minv      <- _test.py:117-184, 186-202, 204-226
    rnea_grad_inner / rnea_grad         <- _test.py:229-488, 490-494
    fd_grad                             <- _test.py:496-520
    rollout_step / rollout              <- no reference counterpart (consumer of forward_dynamics_gradient_device,
                                           README.md:26-29); defined here from fd_grad and minv
    mxS / fxv / fx                      <- _test.py:522-664

Differences from the reference text (behaviour identical unless stated):
  * every function is vectorised over a leading batch axis K (the reference handles one
    configuration per call with Python loops over rows); joints/columns are still looped;
  * X_j(q) is evaluated from the robot object's numeric callables (``get_Xmat_Func_by_id``) through
    an exact fit X = A sin q + B cos q + D q + C, so a batch costs one broadcast, not K calls;
  * ``prismatic_fix``: the reference differentiates the backward pass with ``-X^T mxS(S, f)``
    (_test.py:311,437), i.e. it applies the *motion* cross product to a force vector.  That equals
    the correct ``X^T (S x* f)`` only for revolute joints; for prismatic joints the reference
    gradient disagrees with finite differences (tests/test_oracle.py demonstrates it).  With
    ``prismatic_fix=False`` this oracle reproduces the reference bit-for-bit-in-structure (used to
    pin it against the golden fixtures); with ``True`` (default) it uses the force cross product.
    The two are identical for revolute-only robots (iiwa7, atlas30).

Parity pin: tests/test_oracle.py checks this file against tests/golden/*.npz, which were produced
by importing the reference itself in the build container (tests/golden/make_golden.py).

Conventions: spatial vectors are [omega; v]; gravity enters as the base
acceleration [0,0,0,0,0,g] with g = +9.81 for the reference default GRAVITY=-9.81
(_test.py:13-14: gravity_vec[5] = -GRAVITY); Minv[r, c]; dc_du = hstack(dc_dq, dc_dqd).
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# spatial algebra (batched; trailing axis of vectors is the 6-axis unless noted)
# ----------------------------------------------------------------------------------------------
def mxS(S_ind, vec, alpha=1.0):
    """crm(vec) @ S * alpha for S = e_{S_ind}  (_test.py:522-608).  vec: (..., 6)."""
    out = np.zeros_like(vec)
    a = alpha
    if S_ind == 0:
        out[..., 1] = vec[..., 2] * a; out[..., 2] = -vec[..., 1] * a
        out[..., 4] = vec[..., 5] * a; out[..., 5] = -vec[..., 4] * a
    elif S_ind == 1:
        out[..., 0] = -vec[..., 2] * a; out[..., 2] = vec[..., 0] * a
        out[..., 3] = -vec[..., 5] * a; out[..., 5] = vec[..., 3] * a
    elif S_ind == 2:
        out[..., 0] = vec[..., 1] * a; out[..., 1] = -vec[..., 0] * a
        out[..., 3] = vec[..., 4] * a; out[..., 4] = -vec[..., 3] * a
    elif S_ind == 3:
        out[..., 4] = vec[..., 2] * a; out[..., 5] = -vec[..., 1] * a
    elif S_ind == 4:
        out[..., 3] = -vec[..., 2] * a; out[..., 5] = vec[..., 0] * a
    elif S_ind == 5:
        out[..., 3] = vec[..., 1] * a; out[..., 4] = -vec[..., 0] * a
    return out


def fx(vec):
    """Force cross-product matrix crf(vec)  (_test.py:616-647).  vec: (..., 6) -> (..., 6, 6)."""
    r = np.zeros(vec.shape + (6,), dtype=vec.dtype)
    v = vec
    r[..., 0, 1] = -v[..., 2]; r[..., 0, 2] = v[..., 1]; r[..., 0, 4] = -v[..., 5]; r[..., 0, 5] = v[..., 4]
    r[..., 1, 0] = v[..., 2]; r[..., 1, 2] = -v[..., 0]; r[..., 1, 3] = v[..., 5]; r[..., 1, 5] = -v[..., 3]
    r[..., 2, 0] = -v[..., 1]; r[..., 2, 1] = v[..., 0]; r[..., 2, 3] = -v[..., 4]; r[..., 2, 4] = v[..., 3]
    r[..., 3, 4] = -v[..., 2]; r[..., 3, 5] = v[..., 1]
    r[..., 4, 3] = v[..., 2]; r[..., 4, 5] = -v[..., 0]
    r[..., 5, 3] = -v[..., 1]; r[..., 5, 4] = v[..., 0]
    return r


def fxv(a, b):
    """crf(a) @ b  (_test.py:649-664)."""
    return np.einsum("...ij,...j->...i", fx(a), b)


def fxS(S_ind, vec):
    """crf(S) @ vec for S = e_{S_ind} -- the force cross product the backward derivative needs."""
    S = np.zeros(6)
    S[S_ind] = 1.0
    return np.einsum("ij,...j->...i", fx(S), vec)


# ----------------------------------------------------------------------------------------------
# robot adaptor
# ----------------------------------------------------------------------------------------------
class RobotTables:
    """Plain-array view of a URDFParser-style robot object."""

    def __init__(self, robot):
        self.robot = robot
        n = robot.get_num_pos()
        self.n = n
        self.parent = [robot.get_parent_id(j) for j in range(n)]
        self.S_ind = [int(np.asarray(robot.get_S_by_id(j)).tolist().index(1)) for j in range(n)]
        self.levels = [list(robot.get_ids_by_bfs_level(l)) for l in range(robot.get_max_bfs_level() + 1)]
        self.ancestors = [list(robot.get_ancestors_by_id(j)) for j in range(n)]
        self.subtree = [list(robot.get_subtree_by_id(j)) for j in range(n)]
        self.damping = np.array([robot.get_damping_by_id(j) for j in range(n)], dtype=np.float64)
        self.Imats = np.stack([np.asarray(robot.get_Imat_by_id(j), dtype=np.float64) for j in range(n)])
        self.basis = [self._fit_basis(robot.get_Xmat_Func_by_id(j)) for j in range(n)]

    @staticmethod
    def _fit_basis(func):
        """Exact fit X(q) = A sin q + B cos q + D q + C from samples of the numeric callable."""
        th = np.array([-2.3, -1.1, -0.4, 0.3, 0.9, 1.7, 2.6])
        Phi = np.stack([np.sin(th), np.cos(th), th, np.ones_like(th)], axis=1)
        Y = np.stack([np.asarray(func(t), dtype=np.float64).reshape(36) for t in th])
        coef, *_ = np.linalg.lstsq(Phi, Y, rcond=None)
        chk = np.array([0.123, -2.9, 3.05])
        for t in chk:
            fit = (np.array([np.sin(t), np.cos(t), t, 1.0]) @ coef).reshape(6, 6)
            if np.abs(fit - np.asarray(func(t))).max() > 1e-10:
                raise ValueError("X(q) is not of the form A sin q + B cos q + D q + C")
        coef[np.abs(coef) < 1e-13] = 0.0
        return coef.reshape(4, 6, 6)

    def Xmats(self, q):
        """q: (K, n) -> X: (K, n, 6, 6)."""
        q = np.asarray(q, dtype=np.float64)
        K = q.shape[0]
        X = np.empty((K, self.n, 6, 6))
        for j in range(self.n):
            A, B, D, C = self.basis[j]
            qj = q[:, j, None, None]
            X[:, j] = A * np.sin(qj) + B * np.cos(qj) + D * qj + C
        return X


def _as_batch(*arrs):
    out = []
    for a in arrs:
        if a is None:
            out.append(None)
            continue
        a = np.asarray(a, dtype=np.float64)
        out.append(a[None, :] if a.ndim == 1 else a)
    return out


def _mv(M, v):
    return np.einsum("...ij,...j->...i", M, v)


def _mtv(M, v):
    return np.einsum("...ji,...j->...i", M, v)


# ----------------------------------------------------------------------------------------------
# RNEA
# ----------------------------------------------------------------------------------------------
def rnea_fpass(T, q, qd, qdd=None, gravity=9.81, X=None):
    """Forward pass by BFS level (_test.py:5-76).  Returns v, a, f of shape (K, n, 6)."""
    q, qd, qdd = _as_batch(q, qd, qdd)
    K, n = qd.shape
    X = T.Xmats(q) if X is None else X
    v = np.zeros((K, n, 6)); a = np.zeros((K, n, 6)); f = np.zeros((K, n, 6))
    gvec = np.zeros(6)
    gvec[5] = gravity  # reference: gravity_vec[5] = -GRAVITY with GRAVITY = -9.81 (_test.py:13-14)
    for level, inds in enumerate(T.levels):
        for j in inds:
            s = T.S_ind[j]
            if level == 0:  # _test.py:25-32
                v[:, j, s] += qd[:, j]
                a[:, j] = _mv(X[:, j], gvec)
                if qdd is not None:
                    a[:, j, s] += qdd[:, j]
            else:  # _test.py:38-59
                p = T.parent[j]
                v[:, j] = _mv(X[:, j], v[:, p])
                a[:, j] = _mv(X[:, j], a[:, p])
                v[:, j, s] += qd[:, j]
                if qdd is not None:
                    a[:, j, s] += qdd[:, j]
                a[:, j] += mxS(s, v[:, j], qd[:, j])
    for j in range(n):  # _test.py:64-67
        Iv = _mv(T.Imats[j], v[:, j])
        Ia = _mv(T.Imats[j], a[:, j])
        f[:, j] = Ia + fxv(v[:, j], Iv)
    return v, a, f


def rnea_bpass(T, q, qd, f, X=None):
    """Backward pass (_test.py:78-107): c_j = S^T f_j, f_parent += X^T f_j, then + damping*qd."""
    q, qd = _as_batch(q, qd)
    K, n = qd.shape
    X = T.Xmats(q) if X is None else X
    c = np.zeros((K, n))
    for level in range(len(T.levels) - 1, -1, -1):
        for j in T.levels[level]:
            c[:, j] = f[:, j, T.S_ind[j]]
            if level != 0:
                f[:, T.parent[j]] += _mtv(X[:, j], f[:, j])
    c += T.damping[None, :] * qd  # _test.py:103-105
    return c, f


def rnea(T, q, qd, qdd=None, gravity=9.81, X=None):
    """(_test.py:109-115) -> c (K,n), v, a, f (K,n,6); f is the *accumulated* force."""
    q, qd, qdd = _as_batch(q, qd, qdd)
    X = T.Xmats(q) if X is None else X
    v, a, f = rnea_fpass(T, q, qd, qdd, gravity, X)
    c, f = rnea_bpass(T, q, qd, f, X)
    return c, v, a, f


# ----------------------------------------------------------------------------------------------
# direct Minv (Carpentier)
# ----------------------------------------------------------------------------------------------
def minv_bpass(T, q, X=None):
    """(_test.py:117-184).  Returns Minv (K,n,n) upper part so far, F (K,n,6,n), U (K,n,6), Dinv (K,n)."""
    (q,) = _as_batch(q)
    K, n = q.shape
    X = T.Xmats(q) if X is None else X
    Minv = np.zeros((K, n, n)); F = np.zeros((K, n, 6, n)); U = np.zeros((K, n, 6)); Dinv = np.zeros((K, n))
    IA = np.broadcast_to(T.Imats, (K, n, 6, 6)).copy()
    for level in range(len(T.levels) - 1, -1, -1):
        inds = T.levels[level]
        for j in inds:  # _test.py:135-145
            s = T.S_ind[j]
            U[:, j] = IA[:, j, :, s]
            Dinv[:, j] = 1.0 / U[:, j, s]
            Minv[:, j, j] = Dinv[:, j]
        for j in inds:  # _test.py:150-154
            s = T.S_ind[j]
            for k in T.subtree[j]:
                Minv[:, j, k] -= Dinv[:, j] * F[:, j, s, k]
        for j in inds:  # _test.py:159-172
            p = T.parent[j]
            if p == -1:
                continue
            for k in T.subtree[j]:
                F[:, j, :, k] += U[:, j] * Minv[:, j, k, None]
                F[:, p, :, k] += _mtv(X[:, j], F[:, j, :, k])
            Ia = IA[:, j] - np.einsum("ki,kj->kij", U[:, j], Dinv[:, j, None] * U[:, j])
            IA[:, p] += np.einsum("kba,kbc,kcd->kad", X[:, j], Ia, X[:, j])
    return Minv, F, U, Dinv


def minv_fpass(T, q, Minv, F, U, Dinv, X=None):
    """(_test.py:186-202) strictly serial over joint ids."""
    (q,) = _as_batch(q)
    K, n = q.shape
    X = T.Xmats(q) if X is None else X
    for j in range(n):
        p = T.parent[j]
        s = T.S_ind[j]
        if p != -1:
            UX = _mtv(X[:, j], U[:, j])  # U^T X
            Minv[:, j, j:] -= Dinv[:, j, None] * np.einsum("ka,kab->kb", UX, F[:, p, :, j:])
        F[:, j, :, j:] = 0.0
        F[:, j, s, j:] = Minv[:, j, j:]
        if p != -1:
            F[:, j, :, j:] += np.einsum("kab,kbc->kac", X[:, j], F[:, p, :, j:])
    return Minv


def densify_minv(Minv):
    """(_test.py:204-211) mirror the upper triangle."""
    up = np.triu(Minv)
    return up + np.swapaxes(np.triu(Minv, 1), -1, -2)


def minv(T, q, output_dense=True, X=None):
    """(_test.py:213-226).  output_dense=False gives what the kernels emit (upper triangle, lower = 0)."""
    (q,) = _as_batch(q)
    X = T.Xmats(q) if X is None else X
    Minv, F, U, Dinv = minv_bpass(T, q, X)
    Minv = minv_fpass(T, q, Minv, F, U, Dinv, X)
    return densify_minv(Minv) if output_dense else np.triu(Minv)


# ----------------------------------------------------------------------------------------------
# gradient of RNEA
# ----------------------------------------------------------------------------------------------
def rnea_grad_inner(T, q, qd, v, a, f, gravity=9.81, X=None, prismatic_fix=True, return_all=False):
    """(_test.py:229-488).  v, a, f: (K, n, 6) with f accumulated.  Returns dc_dq, dc_dqd (K, n, n).

    Arrays are indexed [K, joint, col, 6] (the reference uses [6, col, joint]).
    """
    q, qd = _as_batch(q, qd)
    K, n = qd.shape
    X = T.Xmats(q) if X is None else X
    gvec = np.zeros(6)
    gvec[5] = gravity
    Iv = np.zeros((K, n, 6)); Xv = np.zeros((K, n, 6)); Xa = np.zeros((K, n, 6))
    for j in range(n):  # _test.py:284-295
        p = T.parent[j]
        if p != -1:
            Xv[:, j] = _mv(X[:, j], v[:, p])
            Xa[:, j] = _mv(X[:, j], a[:, p])
        else:
            Xa[:, j] = _mv(X[:, j], gvec)
        Iv[:, j] = _mv(T.Imats[j], v[:, j])
    MxXv = np.zeros((K, n, 6)); MxXa = np.zeros((K, n, 6)); Mxv = np.zeros((K, n, 6)); Mxf = np.zeros((K, n, 6))
    for j in range(n):  # _test.py:306-311
        s = T.S_ind[j]
        MxXv[:, j] = mxS(s, Xv[:, j]); MxXa[:, j] = mxS(s, Xa[:, j]); Mxv[:, j] = mxS(s, v[:, j])
        # reference: Mxf = mxS(S, f) (motion cross product applied to a force; see module docstring)
        Mxf[:, j] = -fxS(s, f[:, j]) if prismatic_fix else mxS(s, f[:, j])

    dv_dq = np.zeros((K, n, n, 6)); dv_dqd = np.zeros((K, n, n, 6))
    da_dq = np.zeros((K, n, n, 6)); da_dqd = np.zeros((K, n, n, 6))
    df_dq = np.zeros((K, n, n, 6)); df_dqd = np.zeros((K, n, n, 6))
    # forward pass: dv/du by level (_test.py:327-344)
    for level, inds in enumerate(T.levels):
        for j in inds:
            p = T.parent[j]
            for col in T.ancestors[j]:
                dv_dq[:, j, col] = _mv(X[:, j], dv_dq[:, p, col])
                dv_dqd[:, j, col] = _mv(X[:, j], dv_dqd[:, p, col])
            if level != 0:
                dv_dq[:, j, j] += MxXv[:, j]
            dv_dqd[:, j, j, T.S_ind[j]] += 1.0
    # da/du = MxS(dv/du)*qd + {MxXa, Mxv} (_test.py:352-362)
    for j in range(n):
        s = T.S_ind[j]
        for col in T.ancestors[j] + [j]:
            da_dq[:, j, col] = mxS(s, dv_dq[:, j, col], qd[:, j])
            da_dqd[:, j, col] = mxS(s, dv_dqd[:, j, col], qd[:, j])
            if col == j:
                da_dq[:, j, col] += MxXa[:, j]
                da_dqd[:, j, col] += Mxv[:, j]
    # da/du += X da_parent/du (_test.py:370-381)
    for level in range(1, len(T.levels)):
        for j in T.levels[level]:
            p = T.parent[j]
            for col in T.ancestors[j] + [j]:
                da_dq[:, j, col] += _mv(X[:, j], da_dq[:, p, col])
                da_dqd[:, j, col] += _mv(X[:, j], da_dqd[:, p, col])
    # df/du = fx(dv/du) Iv + I da/du + (fx(v) I) dv/du (_test.py:389-424)
    for j in range(n):
        Imat = T.Imats[j]
        FxvI = np.einsum("kab,bc->kac", fx(v[:, j]), Imat)
        for col in T.ancestors[j] + [j]:
            df_dq[:, j, col] = fxv(dv_dq[:, j, col], Iv[:, j]) + _mv(Imat, da_dq[:, j, col]) + _mv(FxvI, dv_dq[:, j, col])
            df_dqd[:, j, col] = fxv(dv_dqd[:, j, col], Iv[:, j]) + _mv(Imat, da_dqd[:, j, col]) + _mv(FxvI, dv_dqd[:, j, col])
    Xmxf = np.zeros((K, n, 6))
    for j in range(n):  # _test.py:433-437
        Xmxf[:, j] = -_mtv(X[:, j], Mxf[:, j])
    df_fp_dq = df_dq.copy(); df_fp_dqd = df_dqd.copy()
    # backward pass (_test.py:450-470)
    for level in range(len(T.levels) - 1, 0, -1):
        for j in T.levels[level]:
            p = T.parent[j]
            for col in T.ancestors[j] + T.subtree[j]:
                df_dq[:, p, col] += _mtv(X[:, j], df_dq[:, j, col])
                df_dqd[:, p, col] += _mtv(X[:, j], df_dqd[:, j, col])
                if col == j:
                    df_dq[:, p, col] += Xmxf[:, j]
    # extract (_test.py:479-486)
    dc_dq = np.zeros((K, n, n)); dc_dqd = np.zeros((K, n, n))
    for j in range(n):
        s = T.S_ind[j]
        for col in T.ancestors[j] + T.subtree[j]:
            dc_dq[:, j, col] = df_dq[:, j, col, s]
            dc_dqd[:, j, col] = df_dqd[:, j, col, s] + (T.damping[j] if j == col else 0.0)
    if return_all:
        return dc_dq, dc_dqd, dv_dq, dv_dqd, da_dq, da_dqd, df_fp_dq, df_fp_dqd, df_dq, df_dqd
    return dc_dq, dc_dqd


def rnea_grad(T, q, qd, qdd=None, gravity=9.81, prismatic_fix=True):
    """(_test.py:490-494) -> dc_du (K, n, 2n) = hstack(dc_dq, dc_dqd)."""
    q, qd, qdd = _as_batch(q, qd, qdd)
    X = T.Xmats(q)
    c, v, a, f = rnea(T, q, qd, qdd, gravity, X)
    dc_dq, dc_dqd = rnea_grad_inner(T, q, qd, v, a, f, gravity, X, prismatic_fix)
    return np.concatenate([dc_dq, dc_dqd], axis=2)


def forward_dynamics(T, q, qd, u, gravity=9.81):
    """qdd = Minv (u - c)  (_test.py:498-501)."""
    q, qd, u = _as_batch(q, qd, u)
    X = T.Xmats(q)
    c = rnea(T, q, qd, None, gravity, X)[0]
    Mi = minv(T, q, True, X)
    return np.einsum("kij,kj->ki", Mi, u - c)


def fd_grad(T, q, qd, u, gravity=9.81, prismatic_fix=True, return_parts=False):
    """(_test.py:496-520) -> df_du (K, n, 2n) = -Minv @ dc_du at qdd = Minv (u - c)."""
    q, qd, u = _as_batch(q, qd, u)
    X = T.Xmats(q)
    c = rnea(T, q, qd, None, gravity, X)[0]
    Mi = minv(T, q, True, X)
    qdd = np.einsum("kij,kj->ki", Mi, u - c)
    c2, v, a, f = rnea(T, q, qd, qdd, gravity, X)
    dc_dq, dc_dqd = rnea_grad_inner(T, q, qd, v, a, f, gravity, X, prismatic_fix)
    dc_du = np.concatenate([dc_dq, dc_dqd], axis=2)
    df_du = -np.einsum("kij,kjl->kil", Mi, dc_du)
    if return_parts:
        return df_du, dict(c=c, Minv=Mi, qdd=qdd, dc_du=dc_du, v=v, a=a, f=f)
    return df_du


# ----------------------------------------------------------------------------------------------
# boundary layouts: what the kernels read / write, flattened
# ----------------------------------------------------------------------------------------------
def rollout_step(T, q, qd, u, dt, gravity=9.81):
    """One semi-implicit Euler step of the forward dynamics and its linearisation (the consumer the reference's _device tier
    exists for, README.md:26-29; there is no reference implementation of it -- this is the definition the HIP rollout
    kernel is tested against):

        qdd = FD(q, qd, u);   qd+ = qd + dt qdd;   q+ = q + dt qd+        x = [q; qd]
        A = dx+/dx = [[I + dt^2 dqdd/dq,  dt I + dt^2 dqdd/dqd], [dt dqdd/dq,  I + dt dqdd/dqd]]      (2n x 2n)
        B = dx+/du = [[dt^2 Minv], [dt Minv]]                                                         (2n x n)

    Returns (q+, qd+, A, B), batched."""
    q, qd, u = _as_batch(q, qd, u)
    n = q.shape[1]
    df, parts = fd_grad(T, q, qd, u, gravity, return_parts=True)
    qdd, Minv = parts["qdd"], parts["Minv"]
    qd_next = qd + dt * qdd
    q_next = q + dt * qd_next
    I = np.eye(n)[None]
    dq, dqd = df[:, :, :n], df[:, :, n:]
    A = np.concatenate([np.concatenate([I + dt * dt * dq, dt * I + dt * dt * dqd], axis=2),
                        np.concatenate([dt * dq, I + dt * dqd], axis=2)], axis=1)
    B = np.concatenate([dt * dt * Minv, dt * Minv], axis=1)
    return q_next, qd_next, A, B


def rollout(T, q0, qd0, u_traj, dt, gravity=9.81):
    """T-step rollout from (q0, qd0) under u_traj[t] (shape (steps, K, n)).  Returns x (steps, K, 2n) = the states AFTER each
    step, A (steps, K, 2n, 2n), B (steps, K, 2n, n)."""
    q, qd = _as_batch(q0, qd0)
    xs, As, Bs = [], [], []
    for t in range(u_traj.shape[0]):
        q, qd, A, B = rollout_step(T, q, qd, u_traj[t], dt, gravity)
        xs.append(np.concatenate([q, qd], axis=1)); As.append(A); Bs.append(B)
    return np.stack(xs), np.stack(As), np.stack(Bs)


def pack_q_qd_u(q, qd, u):
    """[K][3n] = [q | qd | u]."""
    return np.concatenate([q, qd, u], axis=1)


def flat_colmajor(M):
    """(K, r, c) -> (K, r*c) column-major (element [r, c] at c*rows + r)."""
    return np.swapaxes(M, 1, 2).reshape(M.shape[0], -1)
