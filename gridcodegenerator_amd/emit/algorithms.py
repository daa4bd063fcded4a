"""Rigid-body-dynamics algorithms written against the tracer (trace.py).

Each function runs the algorithm ONCE at generation time on traced scalars for one configuration --
the emitted code is what a single wavefront lane executes.  The mathematics is the reference's:

    RNEA                 algorithms/_inverse_dynamics.py:33-304     (oracle: _test.py:5-115)
    direct Minv          algorithms/_direct_minv.py:23-382          (oracle: _test.py:117-226)
    FD finish            algorithms/_forward_dynamics.py:21-49
    RNEA gradient        algorithms/_inverse_dynamics_gradient.py:27-650  (oracle: _test.py:229-488)
    FD gradient          algorithms/_forward_dynamics_gradient.py:7-57    (oracle: _test.py:496-520)

but the *schedule* is not: there are no BFS levels, barriers, thread-strided loops or atomics (one
lane owns the whole configuration, so joints are simply visited in id order: parent < child), the
reference's sparsity-compressed column storage (section 8(a) a8/a12 of SURVEY.md) is replaced by
dictionaries keyed (joint, column) that only ever hold the structurally non-zero columns, and every
6x6 product is specialised entry-by-entry by the tracer.
"""
from .trace import P, Tracer, V

_I3 = range(3)


# ------------------------------------------------------------------------------------------------
# spatial algebra on lists of traced scalars
# ------------------------------------------------------------------------------------------------
def zeros6(tr):
    return [tr.zero() for _ in range(6)]


def vadd(a, b):
    return [x + y for x, y in zip(a, b)]


def vsub(a, b):
    return [x - y for x, y in zip(a, b)]


def matvec(tr, M, v):
    return [tr.dot([(M[r][c], v[c]) for c in range(6)]) for r in range(6)]


def matvec_acc(tr, M, v, acc):
    return [tr.dot([(M[r][c], v[c]) for c in range(6)], init=acc[r]) for r in range(6)]


def mattvec(tr, M, v):
    return [tr.dot([(M[r][c], v[r]) for r in range(6)]) for c in range(6)]


def mattvec_acc(tr, M, v, acc):
    return [tr.dot([(M[r][c], v[r]) for r in range(6)], init=acc[c]) for c in range(6)]


def mxS(tr, s, vec, alpha=None):
    """crm(vec) e_s * alpha  (reference helpers/_spatial_algebra_helpers.py:62-147 mxK family)."""
    z = tr.zero()
    v = vec
    table = {
        0: [z, v[2], -v[1], z, v[5], -v[4]],
        1: [-v[2], z, v[0], -v[5], z, v[3]],
        2: [v[1], -v[0], z, v[4], -v[3], z],
        3: [z, z, z, z, v[2], -v[1]],
        4: [z, z, z, -v[2], z, v[0]],
        5: [z, z, z, v[1], -v[0], z],
    }[s]
    if alpha is None:
        return table
    return [x * alpha for x in table]


def fxv(tr, a, b):
    """crf(a) b  (reference helpers/_spatial_algebra_helpers.py:234-257 fx_times_v)."""
    d = tr.dot
    return [
        d([(-a[2], b[1]), (a[1], b[2]), (-a[5], b[4]), (a[4], b[5])]),
        d([(a[2], b[0]), (-a[0], b[2]), (a[5], b[3]), (-a[3], b[5])]),
        d([(-a[1], b[0]), (a[0], b[1]), (-a[4], b[3]), (a[3], b[4])]),
        d([(-a[2], b[4]), (a[1], b[5])]),
        d([(a[2], b[3]), (-a[0], b[5])]),
        d([(-a[1], b[3]), (a[0], b[4])]),
    ]


PRISMATIC_GRADIENT = "corrected"        # module-wide switch, set per build by GRiDCodeGenerator(prismatic_gradient=...)


def fxS(tr, s, f):
    """crf(e_s) f -- force cross product with the joint axis: the seed of the d/dq column of joint s's own joint in the backward pass
    of the RNEA gradient.  The reference uses -mxS(S, f) there, the MOTION cross product applied to a force (_test.py:311,437, emitted
    identically): the same thing for revolute joints (crm(S) is block-diagonal and antisymmetric), different -- and at odds with finite
    differences -- for prismatic ones (tests/test_oracle.py).  PRISMATIC_GRADIENT = "reference" reproduces the reference's choice so
    that a prismatic robot's gradients can be held to the reference's own numbers; "corrected" (default) is the force cross product."""
    if PRISMATIC_GRADIENT == "reference":
        return [-x for x in mxS(tr, s, f)]
    S = [tr.const(1.0 if i == s else 0.0) for i in range(6)]
    return fxv(tr, S, f)


# ------------------------------------------------------------------------------------------------
# model quantities
# ------------------------------------------------------------------------------------------------
def build_X(tr, spec, q, trig):
    """X_j(q_j) entries as traced scalars.  trig[j] = (sin q_j, cos q_j) or None for joints without trig."""
    X = []
    for j in range(spec.n):
        A, B, D, C = spec.Xbasis[j]
        s, c = trig[j] if trig[j] is not None else (tr.zero(), tr.zero())
        Xj = [[None] * 6 for _ in range(6)]
        for r in range(6):
            for col in range(6):
                if r < 3 and col >= 3:
                    Xj[r][col] = tr.zero()
                    continue
                if r >= 3 and col >= 3:  # bottom-right == top-left
                    Xj[r][col] = Xj[r - 3][col - 3]
                    continue
                Xj[r][col] = tr.dot([(float(A[r, col]), s), (float(B[r, col]), c), (float(D[r, col]), q[j])],
                                    init=tr.const(float(C[r, col])))
        X.append(Xj)
    return X


def build_I(tr, spec):
    return [[[tr.const(float(spec.Imats[j][r, c])) for c in range(6)] for r in range(6)] for j in range(spec.n)]


def trig_from_q(tr, spec, q):
    return [(tr.sin(q[j]), tr.cos(q[j])) if spec.uses_trig[j] else None for j in range(spec.n)]


# ------------------------------------------------------------------------------------------------
# RNEA
# ------------------------------------------------------------------------------------------------
def rnea(tr, spec, X, I, qd, qdd, gravity):
    """Returns c (n), v, a, f (n x 6; f accumulated over subtrees as in the reference's s_vaf)."""
    n = spec.n
    v = [None] * n; a = [None] * n; f = [None] * n
    for j in range(n):
        p, s = spec.parent[j], spec.S_ind[j]
        if p == -1:
            vj = zeros6(tr)
            vj[s] = qd[j]
            aj = [X[j][r][5] * gravity for r in range(6)]
        else:
            vj = matvec(tr, X[j], v[p])
            vj[s] = vj[s] + qd[j]
            aj = matvec(tr, X[j], a[p])
            aj = vadd(aj, mxS(tr, s, vj, qd[j]))
        if qdd is not None:
            aj[s] = aj[s] + qdd[j]
        v[j], a[j] = vj, aj
        Iv = matvec(tr, I[j], vj)
        f[j] = vadd(matvec(tr, I[j], aj), fxv(tr, vj, Iv))
    c = [None] * n
    for j in range(n - 1, -1, -1):
        p, s = spec.parent[j], spec.S_ind[j]
        c[j] = f[j][s] + qd[j] * spec.damping[j]
        if p != -1:
            f[p] = mattvec_acc(tr, X[j], f[j], f[p])
    return c, v, a, f


# ------------------------------------------------------------------------------------------------
# direct Minv
# ------------------------------------------------------------------------------------------------
def direct_minv(tr, spec, X, I, between=None, on_final=None):
    """Upper-triangular Minv[j][k] (k >= j) as traced scalars; entries k < j are None.

    between(carried): called between the backward and the forward pass with the backward-pass values the forward pass uses.  on_final(j, k, value): called as soon as Minv[j][k] has its
    final value (row j is final when the forward pass has visited joint j) -- a consumer that uses the entry right there (the
    tile-cooperative producer publishes it and folds it into qdd) lets it die instead of keeping all n(n+1)/2 alive."""
    with tr.mixed_region():
        return _direct_minv(tr, spec, X, I, between, on_final)


def _direct_minv(tr, spec, X, I, between=None, on_final=None):
    n = spec.n
    IA = [[[I[j][r][c] if r <= c else None for c in range(6)] for r in range(6)] for j in range(n)]
    for j in range(n):  # mirror so IA[r][c] is IA[c][r] (symmetric by construction)
        for r in range(6):
            for c in range(r):
                IA[j][r][c] = IA[j][c][r]
    Minv = [[None] * n for _ in range(n)]
    F = {}       # backward-pass F[(j, k)] -> 6-vector
    U = [None] * n
    Dinv = [None] * n
    for j in range(n - 1, -1, -1):
        p, s = spec.parent[j], spec.S_ind[j]
        Uj = [IA[j][r][s] for r in range(6)]
        Dj = tr.rcp(Uj[s])
        U[j], Dinv[j] = Uj, Dj
        Minv[j][j] = Dj
        for k in spec.subtree[j]:
            if k != j:
                Fjk = F.get((j, k))
                Minv[j][k] = -(Dj * Fjk[s]) if Fjk is not None else tr.zero()
        if p == -1:
            continue
        for k in spec.subtree[j]:
            Fjk = F.get((j, k), None)
            upd = [Uj[r] * Minv[j][k] for r in range(6)]
            Fjk = upd if Fjk is None else vadd(Fjk, upd)
            F[(j, k)] = Fjk
            F[(p, k)] = mattvec_acc(tr, X[j], Fjk, F.get((p, k), zeros6(tr)))
        # articulated inertia: Ia = IA - U Dinv U^T (row/col s vanish identically), IA_p += X^T Ia X
        UD = [Uj[r] * Dj for r in range(6)]
        Ia = [[None] * 6 for _ in range(6)]
        for r in range(6):
            for c in range(r, 6):
                if r == s or c == s:
                    val = tr.zero()
                else:
                    val = tr.fma(-UD[r], Uj[c], IA[j][r][c])
                Ia[r][c] = val
                Ia[c][r] = val
        Tm = [[tr.dot([(Ia[r][k], X[j][k][c]) for k in range(6)]) for c in range(6)] for r in range(6)]
        for r in range(6):
            for c in range(r, 6):
                val = tr.dot([(X[j][k][r], Tm[k][c]) for k in range(6)], init=IA[p][r][c])
                IA[p][r][c] = val
                IA[p][c][r] = val
    if between is not None:
        # everything the forward pass still needs from the backward pass (for emitters that must pin it before a barrier)
        carried = [Dinv[j] for j in range(n)] + [U[j][r] for j in range(n) for r in range(6)]
        carried += [Minv[j][k] for j in range(n) for k in range(j, n) if Minv[j][k] is not None]
        between(carried)
    Fn = {}      # forward-pass F (the reference overwrites F in place, _test.py:198-200)
    for j in range(n):
        p, s = spec.parent[j], spec.S_ind[j]
        if p != -1:
            UX = mattvec(tr, X[j], U[j])
            for k in range(j, n):
                Fpk = Fn.get((p, k))
                if Fpk is None:
                    continue
                corr = Dinv[j] * tr.dot([(UX[r], Fpk[r]) for r in range(6)])
                Minv[j][k] = (Minv[j][k] if Minv[j][k] is not None else tr.zero()) - corr
        for k in range(j, n):
            if Minv[j][k] is None:
                Minv[j][k] = tr.zero()
            if on_final is not None:
                on_final(j, k, Minv[j][k])
        if not spec.children[j]:
            continue
        for k in range(j, n):
            Fjk = zeros6(tr)
            Fjk[s] = Minv[j][k]
            if p != -1 and (p, k) in Fn:
                Fjk = matvec_acc(tr, X[j], Fn[(p, k)], Fjk)
            if all(x.is_zero() for x in Fjk):
                continue
            Fn[(j, k)] = Fjk
    return Minv


def minv_backward_lean(tr, spec, I, X_back, publish, joints, cols=None):
    """Backward pass of the Minv recursion (_direct_minv, first loop) over whole base-rooted trees (`joints`), for blocks whose waves
    share the recursion through LDS: every value the forward pass needs is PUBLISHED where it is produced -- publish("U", j, r, v)
    r = 0..5, publish("D", j, 0, 1/D_j), publish("M", j, k, backward-pass value of Minv[j][k]) for k in subtree(j) -- and nothing is
    carried in registers.  X_back(j): X_j(q_j) built from re-read sin / cos.  cols: only these columns' entries and F recursions (two
    waves may divide a large tree's columns; each then repeats the articulated-inertia chain, whose U and 1/D both publish -- the
    same values to the same words)."""
    n = spec.n
    inside = set(joints)
    wanted = (lambda k: True) if cols is None else (lambda k, cs=set(cols): k in cs)
    IA = {}

    def IA_of(j):
        if j not in IA:
            IA[j] = [[I[j][min(r, c)][max(r, c)] for c in range(6)] for r in range(6)]
        return IA[j]
    F = {}
    with tr.mixed_region():
        for j in range(n - 1, -1, -1):
            if j not in inside:
                continue
            p, s = spec.parent[j], spec.S_ind[j]
            IAj = IA_of(j)
            Uj = [IAj[r][s] for r in range(6)]
            Dj = tr.rcp(Uj[s])
            if p != -1:                      # (a base joint's U is never needed)
                for r in range(6):
                    publish("U", j, r, Uj[r])
            publish("D", j, 0, Dj)
            Mj = {j: Dj}
            mine = [k for k in spec.subtree[j] if wanted(k)]
            for k in mine:
                if k != j:
                    Fjk = F.get((j, k))
                    Mj[k] = -(Dj * Fjk[s]) if Fjk is not None else tr.zero()
            for k in mine:
                publish("M", j, k, Mj[k])
            if p == -1:
                continue
            Xj = X_back(j)
            for k in mine:
                Fjk = F.pop((j, k), None)
                upd = [Uj[r] * Mj[k] for r in range(6)]
                Fjk = upd if Fjk is None else vadd(Fjk, upd)
                F[(p, k)] = mattvec_acc(tr, Xj, Fjk, F.get((p, k), zeros6(tr)))
            UD = [Uj[r] * Dj for r in range(6)]
            Ia = [[None] * 6 for _ in range(6)]
            for r in range(6):
                for c in range(r, 6):
                    val = tr.zero() if (r == s or c == s) else tr.fma(-UD[r], Uj[c], IAj[r][c])
                    Ia[r][c] = val
                    Ia[c][r] = val
            Tm = [[tr.dot([(Ia[r][k], Xj[k][c]) for k in range(6)]) for c in range(6)] for r in range(6)]
            IAp = IA_of(p)
            for r in range(6):
                for c in range(r, 6):
                    val = tr.dot([(Xj[k][r], Tm[k][c]) for k in range(6)], init=IAp[r][c])
                    IAp[r][c] = val
                    IAp[c][r] = val
            del IA[j]


def minv_forward_lean(tr, spec, X_fwd, fetch, cols, on_final):
    """Forward pass of the Minv recursion (_direct_minv, second loop) for the columns `cols` only -- the pass is independent per column,
    so the waves of a block divide the columns -- from the published values of minv_backward_lean: fetch("U" | "D" | "M", j, i).
    on_final(j, k, value): Minv[j][k] (k >= j, k in cols) has its final value; not called where the backward-pass value already is
    the final one (base joints: the published slot is right as it stands)."""
    n = spec.n
    cols = sorted(cols)
    tree_root = {}
    for j in range(n):
        tree_root[j] = j if spec.parent[j] == -1 else tree_root[spec.parent[j]]
    roots = set(tree_root[k] for k in cols)
    Fn = {}
    with tr.mixed_region():
        for j in range(n):
            if tree_root[j] not in roots:
                continue
            mine = [k for k in cols if k >= j and tree_root[k] == tree_root[j]]
            if not mine:
                continue
            p, s = spec.parent[j], spec.S_ind[j]
            # joints off every path root -> column contribute only where F of the parent is non-zero: all joints j <= k of the tree do
            need_X = (p != -1) or bool(spec.children[j])
            Xj = X_fwd(j) if need_X else None
            if p != -1:
                UX = mattvec(tr, Xj, [fetch("U", j, r) for r in range(6)])
                Dj = fetch("D", j, 0)
            Mj = {}
            for k in mine:
                val = fetch("M", j, k) if k in spec.subtree[j] else tr.zero()
                if p != -1 and (p, k) in Fn:
                    val = val - Dj * tr.dot([(UX[r], Fn[(p, k)][r]) for r in range(6)])
                    on_final(j, k, val)
                elif p != -1 and k not in spec.subtree[j]:
                    on_final(j, k, val)          # (a structural zero of the forward pass: written so that the slot holds a value)
                Mj[k] = val
            if not spec.children[j]:
                continue
            for k in mine:
                Fjk = zeros6(tr)
                Fjk[s] = Mj[k]
                if p != -1 and (p, k) in Fn:
                    Fjk = matvec_acc(tr, Xj, Fn[(p, k)], Fjk)
                if all(x.is_zero() for x in Fjk):
                    continue
                Fn[(j, k)] = Fjk


def minv_columns_lean(tr, spec, X_of, fetch, cols, on_final):
    """Columns `cols` of Minv from the PUBLISHED articulated-inertia chain alone (minv_backward_lean with cols = [] publishes U_j and
    1/D_j of every joint and nothing else): everything else in the Minv recursion is independent per column, so the waves of a block
    divide it by columns with no further exchange.  For each column k: the backward pass's F recursion from joint k up to the base
    (Minv[j][k] = -F[j][s_j] / D_j for the ancestors j of k; _direct_minv first loop, restricted to the column), then the forward
    pass down every joint j <= k of the tree (second loop).  The backward-pass entries never leave the wave's registers.
    fetch("U" | "D", j, i): the published values; X_of(j): X_j(q_j) built from re-read sin / cos (once per pass);
    on_final(j, k, value): Minv[j][k] (j <= k) has its final value."""
    n = spec.n
    cols = sorted(cols)
    tree_root = {}
    for j in range(n):
        tree_root[j] = j if spec.parent[j] == -1 else tree_root[spec.parent[j]]
    roots = sorted(set(tree_root[k] for k in cols))
    with tr.mixed_region():
        for root in roots:
            mine_all = [k for k in cols if tree_root[k] == root]
            joints = [j for j in range(n) if tree_root[j] == root and j <= mine_all[-1]]
            # ---- backward: F[(j, k)] for the joints j on the path base -> k, from k upwards
            F, M = {}, {}
            for j in reversed(joints):
                mine = [k for k in mine_all if k in spec.subtree[j]]
                if not mine:
                    continue
                p, s = spec.parent[j], spec.S_ind[j]
                Dj = fetch("D", j, 0)
                for k in mine:
                    M[(j, k)] = Dj if k == j else -(Dj * F[(j, k)][s])
                if p == -1:
                    continue
                Uj = [fetch("U", j, r) for r in range(6)]
                Xj = X_of(j)
                for k in mine:
                    Fjk = F.pop((j, k), None)
                    upd = [Uj[r] * M[(j, k)] for r in range(6)]
                    Fjk = upd if Fjk is None else vadd(Fjk, upd)
                    F[(p, k)] = mattvec(tr, Xj, Fjk)
            # ---- forward: every joint j <= k of the tree, in id order
            Fn = {}
            for j in joints:
                mine = [k for k in mine_all if k >= j]
                if not mine:
                    continue
                p, s = spec.parent[j], spec.S_ind[j]
                need_X = (p != -1) or bool(spec.children[j])
                Xj = X_of(j) if need_X else None
                if p != -1:
                    UX = mattvec(tr, Xj, [fetch("U", j, r) for r in range(6)])
                    Dj = fetch("D", j, 0)
                Mj = {}
                for k in mine:
                    val = M.get((j, k), tr.zero())
                    if p != -1 and (p, k) in Fn:
                        val = val - Dj * tr.dot([(UX[r], Fn[(p, k)][r]) for r in range(6)])
                    Mj[k] = val
                    on_final(j, k, val)
                if not spec.children[j]:
                    continue
                for k in mine:
                    Fjk = zeros6(tr)
                    Fjk[s] = Mj[k]
                    if p != -1 and (p, k) in Fn:
                        Fjk = matvec_acc(tr, Xj, Fn[(p, k)], Fjk)
                    if all(x.is_zero() for x in Fjk):
                        continue
                    Fn[(j, k)] = Fjk


def minv_sym(Minv, r, c):
    return Minv[r][c] if r <= c else Minv[c][r]


def fd_finish(tr, spec, Minv, u, c):
    """qdd = Minv_sym (u - c)  (algorithms/_forward_dynamics.py:21-49)."""
    n = spec.n
    with tr.mixed_region():
        umc = [u[k] - c[k] for k in range(n)]
        qdd = [tr.dot([(minv_sym(Minv, r, k), umc[k]) for k in range(n)]) for r in range(n)]
    return [tr.cast(x, 0) if tr.mixed else x for x in qdd]      # everything downstream (RNEA at qdd, the gradient) is in C


def minv_in_compute_type(tr, Minv):
    """Minv entries converted once to the compute type C (mixed precision: the recursion ran in D)."""
    return [[tr.cast(e, 0) if (e is not None and tr.mixed) else e for e in row] for row in Minv]


# ------------------------------------------------------------------------------------------------
# gradient of RNEA
# ------------------------------------------------------------------------------------------------
def rnea_grad(tr, spec, X, I, qd, v, a, f, gravity):
    """dc_du[j][col] as PAIRS (lo = d/dq, hi = d/dqd; structurally-zero entries are const 0).

    v, a, f as returned by rnea() (f accumulated).  Follows _test.py:229-488 column by column; only the columns
    (ancestors + self, then + subtree on the way back) that can be non-zero exist.  The d/dq and d/dqd versions of
    every quantity go through identical operations, so they travel as one pair (trace.P) and are emitted as
    packed instructions.
    """
    n = spec.n
    zero = tr.zero()
    dv = {}; da = {}; df = {}       # (j, col) -> 6-vector of pairs
    for j in range(n):
        p, s = spec.parent[j], spec.S_ind[j]
        if p != -1:
            Xv = matvec(tr, X[j], v[p])
            Xa = matvec(tr, X[j], a[p])
        else:
            Xv = None
            Xa = [X[j][r][5] * gravity for r in range(6)]
        Iv = matvec(tr, I[j], v[j])
        for col in spec.ancestors[j] + [j]:
            if col != j:
                dvp = matvec(tr, X[j], dv[(p, col)])
            else:
                seed_q = mxS(tr, s, Xv) if Xv is not None else zeros6(tr)
                dvp = [P(tr, seed_q[r], 1.0 if r == s else 0.0) for r in range(6)]
            dv[(j, col)] = dvp
            dap = mxS(tr, s, dvp, qd[j])
            if col == j:
                sq, sqd = mxS(tr, s, Xa), mxS(tr, s, v[j])
                dap = [dap[r] + P(tr, sq[r], sqd[r]) for r in range(6)]
            elif p != -1:
                dap = matvec_acc(tr, X[j], da[(p, col)], dap)
            da[(j, col)] = dap
            # df = I da + crf(v) (I dv) + crf(dv) (I v)
            Idv = matvec(tr, I[j], dvp)
            t = matvec(tr, I[j], dap)
            t = vadd(t, fxv(tr, v[j], Idv))
            t = vadd(t, fxv(tr, dvp, Iv))
            df[(j, col)] = t
    # backward pass: df_parent[col] += X^T df_j[col] (+ X^T crf(S) f_j on the self column of d/dq)
    for j in range(n - 1, 0, -1):
        p, s = spec.parent[j], spec.S_ind[j]
        if p == -1:
            continue
        seed = fxS(tr, s, f[j])
        for col in spec.ancestors[j] + spec.subtree[j]:
            dfj = df[(j, col)]
            if col == j:
                dfj = [dfj[r] + P(tr, seed[r], 0.0) for r in range(6)]
            prev = df.get((p, col), [zero] * 6)
            df[(p, col)] = mattvec_acc(tr, X[j], dfj, prev)
    dc = [[P(tr, 0.0, 0.0) for _ in range(n)] for _ in range(n)]
    for j in range(n):
        s = spec.S_ind[j]
        for col in spec.ancestors[j] + spec.subtree[j]:
            e = df[(j, col)][s]
            e = e if isinstance(e, P) else P(tr, e, e)
            if col == j and spec.damping[j] != 0.0:
                e = P(tr, e.lo, e.hi + spec.damping[j])
            dc[j][col] = e
    return dc


def sym_minv_times_columns(tr, spec, entry, dc_lo, dc_hi, block=16):
    """-Minv_sym @ [dc_lo | dc_hi] for ONE gradient column with every upper-triangle entry of Minv fetched once:
    entry(r, k) (r <= k) -> traced value or None (structural zero); dc_lo / dc_hi: {row: value} of the non-zero rows.
    Up to 4 multiply-adds per fetched entry (both halves, both triangles) -- for cores that re-read Minv from LDS per column.
    The fetches are software-pipelined in trace order: the entries of block b + 1 are requested before the multiply-adds of
    block b (creation-order emission keeps that order), so a lone wavefront does not pay an LDS round trip per entry."""
    n = spec.n
    acc_lo = [tr.zero() for _ in range(n)]
    acc_hi = [tr.zero() for _ in range(n)]
    need = []
    for k in range(n):
        for r in range(k + 1):
            need_rk = k in dc_lo             # contributes to row r through dc[k]
            need_kr = (r != k) and (r in dc_lo)
            if need_rk or need_kr:
                need.append((r, k, need_rk, need_kr))
    blocks = [need[i:i + block] for i in range(0, len(need), block)]
    fetch = lambda blk: [entry(r, k) for (r, k, _, _) in blk]
    pending = fetch(blocks[0]) if blocks else []
    for b, blk in enumerate(blocks):
        vals = pending
        pending = fetch(blocks[b + 1]) if b + 1 < len(blocks) else []
        for (r, k, need_rk, need_kr), m in zip(blk, vals):
            if m is None:
                continue
            if need_rk:
                acc_lo[r] = tr.fma(m, dc_lo[k], acc_lo[r]); acc_hi[r] = tr.fma(m, dc_hi[k], acc_hi[r])
            if need_kr:
                acc_lo[k] = tr.fma(m, dc_lo[r], acc_lo[k]); acc_hi[k] = tr.fma(m, dc_hi[r], acc_hi[k])
    return [-x for x in acc_lo], [-x for x in acc_hi]


def sym_minv_times_column_pair(tr, spec, entry, dc_a, dc_b, block=16):
    """-Minv_sym @ dc for TWO gradient half-columns of one base-rooted tree at once: every upper-triangle entry of Minv is fetched
    once for both (entry(r, k), r <= k, as in sym_minv_times_columns) and each multiply-add serves both columns as ONE packed
    instruction (v_pk_fma_f32: accumulator pair += entry x (dc_a[k], dc_b[k])) -- half the LDS reads and half the vector
    instructions of two separate products, in a kernel that is bound by vector-instruction issue.  Rows that only one column has
    stay scalar (the pair folds).  dc_a / dc_b: {row: value}.  Returns (column a, column b), n values each."""
    n = spec.n
    zero = tr.zero()
    rows = set(dc_a) | set(dc_b)
    pair = {k: P(tr, dc_a.get(k, zero), dc_b.get(k, zero)) for k in rows}
    acc = [P(tr, zero, zero) for _ in range(n)]
    need = []
    for k in range(n):
        for r in range(k + 1):
            need_rk = k in rows
            need_kr = (r != k) and (r in rows)
            if need_rk or need_kr:
                need.append((r, k, need_rk, need_kr))
    blocks = [need[i:i + block] for i in range(0, len(need), block)]
    fetch = lambda blk: [entry(r, k) for (r, k, _, _) in blk]
    pending = fetch(blocks[0]) if blocks else []
    was = Tracer.use_packed
    Tracer.use_packed = True
    try:
        for b, blk in enumerate(blocks):
            vals = pending
            pending = fetch(blocks[b + 1]) if b + 1 < len(blocks) else []
            for (r, k, need_rk, need_kr), m in zip(blk, vals):
                if m is None:
                    continue
                if need_rk:
                    acc[r] = tr.fma(m, pair[k], acc[r])
                if need_kr:
                    acc[k] = tr.fma(m, pair[r], acc[k])
    finally:
        Tracer.use_packed = was
    return [-x.lo for x in acc], [-x.hi for x in acc]


def fd_grad_finish(tr, spec, Minv, dc, cols=None):
    """df_du = -Minv_sym [dc_dq | dc_dqd]  (algorithms/_forward_dynamics_gradient.py:48-57) on pairs; only `cols` if given."""
    n = spec.n
    Minv = minv_in_compute_type(tr, Minv)
    out = [[None] * n for _ in range(n)]
    for col in (range(n) if cols is None else cols):
        for r in range(n):
            e = -tr.dot([(minv_sym(Minv, r, k), dc[k][col]) for k in range(n)])
            out[r][col] = e if isinstance(e, P) else P(tr, e, e)
    return out


# ------------------------------------------------------------------------------------------------
# explicit column-serial schedule for large robots (creation-order emission)
# ------------------------------------------------------------------------------------------------
def build_X_joint(tr, spec, j, qj, trig_j):
    """X_j(q_j) for ONE joint (rematerialised per use-scope by the column-serial schedule)."""
    A, B, D, C = spec.Xbasis[j]
    s, c = trig_j if trig_j is not None else (tr.zero(), tr.zero())
    Xj = [[None] * 6 for _ in range(6)]
    for r in range(6):
        for col in range(6):
            if r < 3 and col >= 3:
                Xj[r][col] = tr.zero()
            elif r >= 3 and col >= 3:
                Xj[r][col] = Xj[r - 3][col - 3]
            else:
                Xj[r][col] = tr.dot([(float(A[r, col]), s), (float(B[r, col]), c), (float(D[r, col]), qj)],
                                    init=tr.const(float(C[r, col])))
    return Xj


def minv_zero_pattern(spec):
    """Structural zeros of the (symmetric) Minv: nz[r][k] is False when Minv[r][k] folds to 0 for every q
    (joints in different base-rooted trees).  Found by tracing direct_minv once and looking for constant entries."""
    from .trace import Tracer
    t = Tracer()
    q = [t.inp("q%d" % j) for j in range(spec.n)]
    X = build_X(t, spec, q, trig_from_q(t, spec, q))
    M = direct_minv(t, spec, X, build_I(t, spec))
    n = spec.n
    return [[not minv_sym(M, r, k).is_zero() for k in range(n)] for r in range(n)]


def rnea_grad_columns(tr, spec, I, q, qd, trig, loader, emit_column, order=None, prefetch=3, xof=None, keep=(), xof_back=None,
                      xa_first=False):
    """Column-serial analytical gradient of RNEA (same mathematics as rnea_grad / _test.py:229-488).

    Designed for robots whose gradient working set does not fit the register file (Atlas-30: 880 live values in the
    demand-ordered trace): v_j, X_j a_parent and the accumulated f_j are NOT kept live -- they are re-read from a
    workspace -- X_j(q) and I_j v_j are rematerialised per column, and a column's forward / backward recursion runs
    depth-first so only one root-to-leaf path of dv, da, df is alive.  The workspace reads are software-pipelined: the
    whole visit sequence is known at generation time, so the loads of visit i + `prefetch` are issued at the start of
    visit i (loader(kind, j) -> 6 fresh load nodes; kinds "v", "xa", "f").  For every column `emit_column(col, dc)`
    receives dc = {row: (d/dq value, d/dqd value)} for the structurally non-zero rows and emits what depends on it.

    xof_back(j): X_j for the way BACK up the tree (child -> parent force transfer); None: the X used on the way down (alive across
    the child's whole subtree, ~12 registers per level of the path).  xa_first: request X_j a_parent of the column's joint before
    its v (the chain root -> joint is then traced joint by joint: X_j, a_j, v_j -- instead of all v first, all a after, which keeps
    every X_j of the path alive in between).
    """
    n = spec.n
    cols = list(order if order is not None else range(n))
    # static plan of load groups in execution order: (column, joint) visits in DFS pre-order of each column's subtree
    plan = []
    for col in cols:
        for j in spec.subtree[col]:
            plan.append((col, j))
    issued = {}
    state = {"next": 0, "pos": 0}

    def issue_upto(limit):
        while state["next"] < min(limit, len(plan)):
            col, j = plan[state["next"]]
            vals = {}
            if j == col and xa_first:
                vals["xa"] = loader("xa", j)
            vals["v"] = loader("v", j)
            if j == col:
                if not xa_first:
                    vals["xa"] = loader("xa", j)
                if spec.parent[j] != -1:
                    vals["f"] = loader("f", j)
            issued[(col, j)] = vals
            state["next"] += 1

    for col in cols:
        mark = tr.cse_mark()
        Xc = {}

        def Xof(j):
            if xof is not None:          # the caller owns the per-column X cache (recompute cores share it with v/a/f)
                return xof(j)
            if j not in Xc:
                Xc[j] = build_X_joint(tr, spec, j, q[j], trig[j])
            return Xc[j]

        dc = {}

        def visit(j, dv_p, da_p):
            s, p = spec.S_ind[j], spec.parent[j]
            tr.fence()                       # loads below are pinned after this point (see grid_in_ws::sync)
            issue_upto(state["pos"] + 1 + prefetch)
            state["pos"] += 1
            got = issued.pop((col, j))
            vj = got["v"]
            Iv = matvec(tr, I[j], vj)
            if j == col:
                if p != -1:
                    Xv = list(vj)
                    Xv[s] = vj[s] - qd[j]
                    dvq = mxS(tr, s, Xv)
                else:
                    dvq = zeros6(tr)
                dvqd = zeros6(tr)
                dvqd[s] = tr.const(1.0)
                daq = vadd(mxS(tr, s, dvq, qd[j]), mxS(tr, s, got["xa"]))
                daqd = vadd(mxS(tr, s, dvqd, qd[j]), mxS(tr, s, vj))
            else:
                X = Xof(j)
                dvq = matvec(tr, X, dv_p[0]); dvqd = matvec(tr, X, dv_p[1])
                daq = matvec_acc(tr, X, da_p[0], mxS(tr, s, dvq, qd[j]))
                daqd = matvec_acc(tr, X, da_p[1], mxS(tr, s, dvqd, qd[j]))
            dfs = []
            for (dvx, dax) in ((dvq, daq), (dvqd, daqd)):
                t = matvec(tr, I[j], dax)
                t = vadd(t, fxv(tr, vj, matvec(tr, I[j], dvx)))
                t = vadd(t, fxv(tr, dvx, Iv))
                dfs.append(t)
            for c in spec.children[j]:
                dfc = visit(c, (dvq, dvqd), (daq, daqd))
                Xch = (xof_back or Xof)(c)
                dfs = [mattvec_acc(tr, Xch, dfc[0], dfs[0]), mattvec_acc(tr, Xch, dfc[1], dfs[1])]
            if j == col and p != -1:
                dfs[0] = vadd(dfs[0], fxS(tr, s, got["f"]))
            dc[j] = (dfs[0][s], dfs[1][s] + spec.damping[j] if (j == col and spec.damping[j] != 0.0) else dfs[1][s])
            return dfs

        cur = visit(col, None, None)
        c = col
        while spec.parent[c] != -1:              # carry the column up through its ancestors
            k = spec.parent[c]
            Xch = (xof_back or Xof)(c)
            cur = [mattvec(tr, Xch, cur[0]), mattvec(tr, Xch, cur[1])]
            dc[k] = (cur[0][spec.S_ind[k]], cur[1][spec.S_ind[k]])
            c = k
        emit_column(col, dc)
        tr.cse_release(mark, keep=keep)
