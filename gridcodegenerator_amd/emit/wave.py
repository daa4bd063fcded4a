"""Wave-per-configuration cores: the lanes of ONE wavefront share a configuration.

The lane-per-configuration kernels need 64 configurations to fill a wave and execute a configuration's whole dependency chain on
one lane (Atlas-30 forward-dynamics gradient: 48 k instructions, 57-59 us however small the batch).  The reference gives a whole
thread block to one configuration and lets its threads split the 6x6 products and the gradient columns
(helpers/_code_generation_helpers.py:41-55, algorithms/_inverse_dynamics_gradient.py:199-246,501-540).  The mapping here:

    lane l < m        column l of d/dq      (m = joints of the wave's group of base-rooted trees)
    lane m + l        column l of d/dqd
    every lane        Minv column (l mod m): the forward dynamics of a unit torque at that joint

* what does not depend on the column -- X(q), the articulated-inertia recursion, both RNEA passes -- is WAVE-UNIFORM: computed
  from inputs that are broadcast with v_readlane_b32 (tracer: bcast), identical in every lane, and parked in a small LDS table
  (v_j, X_j a_parent, accumulated f_j: 18 m words) from which the gradient pass re-reads it;
* what does depend on the column is lane-private: the F recursions of the Minv algorithm (reference algorithms/_direct_minv.py:
  157-163, 311-353 -- `for k in subtree` / `for k >= j` become "every lane its own k"), qdd_k = sum_j Minv[j][k] (u_j - c_j), and the
  dRNEA recursion, one depth-first walk over ALL joints in which a lane's dv / da / df are structurally zero until the walk
  reaches its column's joint, where a per-lane 0/1 mask injects the seeds (reference _inverse_dynamics_gradient.py:240-246,
  282-289 select them by thread index);
* Minv then crosses lanes through LDS (lane k publishes its column, every lane reads every entry as a broadcast) for
  df_du[:, col] = -Minv dc_du[:, col], and the result leaves through an LDS transpose as contiguous runs of the output row.

Base-rooted trees do not interact (block-diagonal Minv, gradients), so each wave of a block takes a run of consecutive trees
(model.SubForest) and the waves of a block never communicate: Atlas-30 = torso + arms + neck (18 joints) | both legs (12).

The lower triangle of Minv comes for free: the reference's forward pass stops at k >= j and relies on symmetry; a lane that
runs it over every joint j obtains the full column k (it is the acceleration response of the whole tree to a unit torque at k).
"""
from . import algorithms as alg
from .model import SubForest, base_trees
from .trace import Tracer

WAVE = 64


def wave_groups(spec, max_waves=4):
    """Runs of consecutive base-rooted trees, one per wave: the largest tree sets the critical path, the others are packed into
    as few further waves as stay below it.  Returns [(first joint, joint count)] or None when a group exceeds 32 joints."""
    trees = base_trees(spec)
    cap = max(c for (_, c) in trees)
    groups = []
    for (first, count) in trees:
        if groups and groups[-1][1] + count <= cap and count < cap:
            groups[-1] = (groups[-1][0], groups[-1][1] + count)
        else:
            groups.append((first, count))
    while len(groups) > max_waves:          # merge the two smallest neighbours
        i = min(range(len(groups) - 1), key=lambda i: groups[i][1] + groups[i + 1][1])
        groups[i:i + 2] = [(groups[i][0], groups[i][1] + groups[i + 1][1])]
    if any(2 * c > WAVE for (_, c) in groups):
        return None
    return groups


class WaveTable:
    """Slots of the wave's uniform LDS table."""

    def __init__(self, m):
        self.V, self.XA, self.F = 0, 6 * m, 12 * m
        self.count = 18 * m


def core_forward_dynamics_gradient_wave(sub):
    """Forward-dynamics gradient of ONE configuration on one wavefront, for the sub-forest `sub` (model.SubForest or a whole
    RobotSpec).  Inputs: in.lane_q/qd/u() = q, qd, u of joint (lane mod m); masks in.mask_k(j) = [lane mod m == j],
    in.mask_dq(j) = [lane == j], in.mask_dqd(j) = [lane == m + j].  Outputs put(r, value): row r of the lane's gradient column."""
    m = sub.n
    tr = Tracer()
    tab = WaveTable(m)
    serial = [0]

    def fresh(expr):
        serial[0] += 1
        return tr.inp("%s/*%d*/" % (expr, serial[0]))

    ql, qdl, ul = tr.inp("in.lane_q()"), tr.inp("in.lane_qd()"), tr.inp("in.lane_u()")
    g = tr.inp("gravity")
    sl = tr.sin(ql) if any(sub.uses_trig) else None
    cl = tr.cos(ql) if any(sub.uses_trig) else None
    I = alg.build_I(tr, sub)

    def uniform_inputs():
        """q, qd, trig of every joint as wave-uniform values (fresh broadcasts: their live ranges start at the request)."""
        q = [tr.bcast(ql, j) for j in range(m)]
        trig = [(tr.bcast(sl, j), tr.bcast(cl, j)) if sub.uses_trig[j] else None for j in range(m)]
        return q, trig

    def X_of(j, q, trig):
        return alg.build_X_joint(tr, sub, j, q[j], trig[j])

    # ---- phase 1: articulated-inertia recursion (uniform) + the Minv recursions with lane = column k ------------------------------
    mark = tr.cse_mark()
    q, trig = uniform_inputs()
    X = [X_of(j, q, trig) for j in range(m)]
    with tr.mixed_region():
        IA = [[[I[j][r][c] for c in range(6)] for r in range(6)] for j in range(m)]
        U, Dinv = [None] * m, [None] * m
        Mcol = [None] * m                      # Mcol[j]: entry [j][k] of Minv in lane k
        F = {}                                 # joint -> per-lane 6-vector: sum over the children c of X_c^T (F_c + U_c Minv[c][k])
        for j in range(m - 1, -1, -1):
            p, s = sub.parent[j], sub.S_ind[j]
            Uj = [IA[j][r][s] for r in range(6)]
            Dj = tr.rcp(Uj[s])
            U[j], Dinv[j] = Uj, Dj
            Mjk = Dj * fresh("in.mask_k(%d)" % j)                       # D_j in the lane that owns column j ...
            if j in F:
                Mjk = Mjk - Dj * F[j][s]                                # ... -D_j F_j[s] in the lanes of its subtree, 0 elsewhere
            Mcol[j] = Mjk
            if p == -1:
                continue
            Fjk = [Uj[r] * Mjk for r in range(6)]
            if j in F:
                Fjk = alg.vadd(F[j], Fjk)
            F[p] = alg.mattvec_acc(tr, X[j], Fjk, F.get(p, alg.zeros6(tr)))
            UD = [Uj[r] * Dj for r in range(6)]
            Ia = [[None] * 6 for _ in range(6)]
            for r in range(6):
                for c in range(r, 6):
                    val = tr.zero() if (r == s or c == s) else tr.fma(-UD[r], Uj[c], IA[j][r][c])
                    Ia[r][c] = val
                    Ia[c][r] = val
            Tm = [[tr.dot([(Ia[r][k], X[j][k][c]) for k in range(6)]) for c in range(6)] for r in range(6)]
            for r in range(6):
                for c in range(r, 6):
                    val = tr.dot([(X[j][k][r], Tm[k][c]) for k in range(6)], init=IA[p][r][c])
                    IA[p][r][c] = val
                    IA[p][c][r] = val
        Fn = {}                                # forward pass: the acceleration of joint j caused by a unit torque at the lane's joint
        for j in range(m):
            p, s = sub.parent[j], sub.S_ind[j]
            if p != -1 and p in Fn:
                UX = alg.mattvec(tr, X[j], U[j])
                Mcol[j] = Mcol[j] - Dinv[j] * tr.dot([(UX[r], Fn[p][r]) for r in range(6)])
            if sub.children[j]:
                Fj = alg.zeros6(tr)
                Fj[s] = Mcol[j]
                if p != -1 and p in Fn:
                    Fj = alg.matvec_acc(tr, X[j], Fn[p], Fj)
                Fn[j] = Fj
    Mc = [tr.cast(e, 0) if tr.mixed else e for e in Mcol]
    for j in range(m):
        tr.m_put(j, Mc[j])
    tr.wave_sync()
    tr.fence()
    keep_m = [e.ref for e in Mcol if not isinstance(e.ref, float)]
    tr.cse_release(mark, keep=keep_m)

    # ---- phases 2 + 3: RNEA at qdd = 0 (bias forces c), qdd, RNEA at qdd; v, X a_parent and the accumulated f go to the table --------
    def rnea_pass(qdd, first):
        mk = tr.cse_mark()
        q, trig = uniform_inputs()
        qd = [tr.bcast(qdl, j) for j in range(m)]
        a = [None] * m
        f = [None] * m
        for j in range(m):
            p, s = sub.parent[j], sub.S_ind[j]
            Xj = X_of(j, q, trig)
            if first:
                if p == -1:
                    vj = alg.zeros6(tr)
                    vj[s] = qd[j]
                else:
                    vj = alg.matvec(tr, Xj, [tr.utab_get(tab.V + 6 * p + r) for r in range(6)])
                    vj[s] = vj[s] + qd[j]
                for r in range(6):
                    tr.utab_put(tab.V + 6 * j + r, vj[r])
            else:
                vj = [tr.utab_get(tab.V + 6 * j + r) for r in range(6)]
            xa = alg.matvec(tr, Xj, a[p]) if p != -1 else [Xj[r][5] * g for r in range(6)]
            aj = list(xa)
            if p != -1:
                aj = alg.vadd(aj, alg.mxS(tr, s, vj, qd[j]))
            if qdd is not None:
                aj[s] = aj[s] + qdd[j]
            a[j] = aj
            if not first:
                for r in range(6):
                    tr.utab_put(tab.XA + 6 * j + r, xa[r])
            Iv = alg.matvec(tr, I[j], vj)
            f[j] = alg.vadd(alg.matvec(tr, I[j], aj), alg.fxv(tr, vj, Iv))
        c = [None] * m
        for j in range(m - 1, -1, -1):
            p, s = sub.parent[j], sub.S_ind[j]
            c[j] = f[j][s] + qd[j] * sub.damping[j]
            if not first:
                for r in range(6):
                    tr.utab_put(tab.F + 6 * j + r, f[j][r])
            if p != -1:
                f[p] = alg.mattvec_acc(tr, X_of(j, q, trig), f[j], f[p])
        return c, mk

    c, mk = rnea_pass(None, True)
    with tr.mixed_region():
        umc = [tr.bcast(ul, j) - c[j] for j in range(m)]
        qdd_lane = tr.dot([(Mcol[j], umc[j]) for j in range(m)])             # lane k: qdd_k = sum_j Minv[j][k] (u_j - c_j)
    qdd_lane = tr.cast(qdd_lane, 0) if tr.mixed else qdd_lane
    tr.anchor(qdd_lane)
    tr.fence()
    tr.cse_release(mark)                      # (Minv columns are in LDS, the bias forces are folded into qdd: nothing else survives)
    qdd = [tr.bcast(qdd_lane, j) for j in range(m)]
    _, mk = rnea_pass(qdd, False)
    tr.fence()
    tr.cse_release(mk)

    # ---- phase 4: dRNEA, lane = column, ONE depth-first walk over all joints ------------------------------------------------------
    dc = [tr.zero()] * m

    def visit(j, dv_p, da_p):
        mk = tr.cse_mark()
        p, s = sub.parent[j], sub.S_ind[j]
        tr.fence()
        qj = tr.bcast(ql, j)
        qdj = tr.bcast(qdl, j)
        trj = (tr.bcast(sl, j), tr.bcast(cl, j)) if sub.uses_trig[j] else None
        Xj = alg.build_X_joint(tr, sub, j, qj, trj)
        vj = [tr.utab_get(tab.V + 6 * j + r) for r in range(6)]
        xa = [tr.utab_get(tab.XA + 6 * j + r) for r in range(6)]
        mq, mqd = fresh("in.mask_dq(%d)" % j), fresh("in.mask_dqd(%d)" % j)
        if p != -1:
            dv = alg.matvec(tr, Xj, dv_p)
            Xv = list(vj)
            Xv[s] = vj[s] - qdj
            seed = alg.mxS(tr, s, Xv)
            dv = [tr.fma(mq, seed[r], dv[r]) for r in range(6)]
        else:
            dv = alg.zeros6(tr)
        dv[s] = dv[s] + mqd
        da = alg.mxS(tr, s, dv, qdj)
        if p != -1:
            da = alg.matvec_acc(tr, Xj, da_p, da)
        sq, sqd = alg.mxS(tr, s, xa), alg.mxS(tr, s, vj)
        da = [tr.fma(mqd, sqd[r], tr.fma(mq, sq[r], da[r])) for r in range(6)]
        Iv = alg.matvec(tr, I[j], vj)
        df = alg.matvec(tr, I[j], da)
        df = alg.vadd(df, alg.fxv(tr, vj, alg.matvec(tr, I[j], dv)))
        df = alg.vadd(df, alg.fxv(tr, dv, Iv))
        keep = [x.ref for x in dv + da + df if not isinstance(x.ref, float)]
        tr.cse_release(mk, keep=keep)
        for ch in sub.children[j]:
            dfc = visit(ch, dv, da)
            mk2 = tr.cse_mark()
            tr.fence()
            Xc = alg.build_X_joint(tr, sub, ch, tr.bcast(ql, ch), (tr.bcast(sl, ch), tr.bcast(cl, ch)) if sub.uses_trig[ch] else None)
            df = alg.mattvec_acc(tr, Xc, dfc, df)
            tr.cse_release(mk2, keep=[x.ref for x in df if not isinstance(x.ref, float)])
        if p != -1:
            fs = alg.fxS(tr, s, [tr.utab_get(tab.F + 6 * j + r) for r in range(6)])
            mq2 = fresh("in.mask_dq(%d)" % j)
            df = [tr.fma(mq2, fs[r], df[r]) for r in range(6)]
        e = df[s]
        if sub.damping[j] != 0.0:
            e = e + fresh("in.mask_dqd(%d)" % j) * sub.damping[j]
        dc[j] = e
        return df

    for root in [j for j in range(m) if sub.parent[j] == -1]:
        visit(root, None, None)

    # ---- phase 5: df_du[:, col] = -Minv dc_du[:, col]; every Minv entry of the upper triangle read once (two multiply-adds) --------
    tr.fence()
    nz = alg.minv_zero_pattern(sub)
    acc = [tr.zero() for _ in range(m)]
    for r in range(m):
        for k in range(r, m):
            if not nz[r][k]:
                continue
            e = tr.m_get(r, k)
            acc[r] = tr.fma(e, dc[k], acc[r])
            if k != r:
                acc[k] = tr.fma(e, dc[r], acc[k])
    for r in range(m):
        tr.out(r, -acc[r])
    return tr
