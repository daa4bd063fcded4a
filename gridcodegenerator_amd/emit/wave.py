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


def wave_roles(spec, groups):
    """Which wave runs the first RNEA pass of another wave's group: {helper wave: helped wave} (at most one pair).  The wave of the
    largest group is the block's critical path; its first RNEA pass (about a sixth of its instructions) needs nothing from the
    articulated-inertia / Minv recursions that the same wave runs first, so the wave of the smallest group takes it over when that
    still leaves it the shorter of the two (joint counts as the measure: work grows faster than linearly in them)."""
    if len(groups) < 2:
        return {}
    big = max(range(len(groups)), key=lambda i: groups[i][1])
    small = min(range(len(groups)), key=lambda i: groups[i][1])
    if big == small or groups[small][1] * 1.25 > groups[big][1]:
        return {}
    return {small: big}


class WaveTable:
    """Slots of the wave's uniform LDS table."""

    def __init__(self, m):
        self.V, self.XA, self.F, self.A, self.C = 0, 6 * m, 12 * m, 18 * m, 24 * m
        self.count = 25 * m


class _Rnea:
    """The two RNEA passes of a wave core for the joint group `sub`, reading the lane's inputs and writing the group's uniform table
    through the accessor methods with suffix `sfx` ("" = the wave's own group; "2" = ANOTHER wave's group: the helper role).

    The recursions v_j = X_j v_p + S qd_j, a_j = X_j a_p + ... are serial along the tree and wave-uniform.  The per-joint force
    f_j = I_j a_j + v_j x* I_j v_j is NOT a recursion: every lane evaluates it for ITS OWN joint (lane l: joint l mod m), dense, with
    per-lane inertia entries (in.lane_I) on v, a gathered from the table -- 90 instructions for all joints at once instead of 66
    uniform ones per joint -- and the backward accumulation f_p += X_j^T f_j broadcasts lane j's result."""

    def __init__(self, tr, sub, g, sfx=""):
        self.tr, self.sub, self.m, self.g, self.sfx = tr, sub, sub.n, g, sfx
        self.tab = WaveTable(sub.n)
        self.ql, self.qdl = tr.inp("in.lane_q%s()" % sfx), tr.inp("in.lane_qd%s()" % sfx)
        self.sl = tr.sin(self.ql) if any(sub.uses_trig) else None
        self.cl = tr.cos(self.ql) if any(sub.uses_trig) else None
        self.Isym = {}
        e = 0
        for r in range(6):
            for c in range(r, 6):
                self.Isym[(r, c)] = self.Isym[(c, r)] = tr.inp("in.lane_I%s(%d)" % (sfx, e))
                e += 1
        self.own = {}                     # per-lane values that live on: v, I v, v x* I v of the lane's own joint
        self.serial = 0

    def put(self, slot, val):
        self.tr.out("utab%s:%d" % (self.sfx, slot), val)

    def lane_tab(self, base, r):
        self.serial += 1
        return self.tr.inp("in.lane_tab%s(%d,%d)/*%d*/" % (self.sfx, base, r, self.serial))

    def sym_mv(self, x):
        return [self.tr.dot([(self.Isym[(r, c)], x[c]) for c in range(6)]) for r in range(6)]

    def uniform_inputs(self):
        """q, trig of every joint as wave-uniform values (fresh broadcasts: their live ranges start at the request)."""
        tr, sub = self.tr, self.sub
        q = [tr.bcast(self.ql, j) for j in range(self.m)]
        trig = [(tr.bcast(self.sl, j), tr.bcast(self.cl, j)) if sub.uses_trig[j] else None for j in range(self.m)]
        return q, trig

    def X_of(self, j, q, trig):
        return alg.build_X_joint(self.tr, self.sub, j, q[j], trig[j])

    def own_refs(self):
        return [x.ref for k in self.own for x in self.own[k] if not isinstance(x.ref, float)]

    def own_from_table(self):
        """v, I v, v x* I v of the lane's own joint from the v another wave parked (helped role)."""
        vl = [self.lane_tab(self.tab.V, r) for r in range(6)]
        Ivl = self.sym_mv(vl)
        self.own.update(v=vl, Iv=Ivl, fxv=alg.fxv(self.tr, vl, Ivl))

    def run(self, qdd, first, park_grad=None):
        """One pass; returns the bias forces c (uniform).  first: computes and parks v; else reads it.  Parks a (every pass), and --
        park_grad (default: not first) -- X_j a_parent and the accumulated forces for the gradient walk."""
        park_grad = (not first) if park_grad is None else park_grad
        tr, sub, m, tab, g = self.tr, self.sub, self.m, self.tab, self.g
        q, trig = self.uniform_inputs()
        qd = [tr.bcast(self.qdl, j) for j in range(m)]
        v, a = [None] * m, [None] * m
        for j in range(m):
            p, s = sub.parent[j], sub.S_ind[j]
            Xj = self.X_of(j, q, trig)
            if first:
                if p == -1:
                    vj = alg.zeros6(tr)
                    vj[s] = qd[j]
                else:
                    vj = alg.matvec(tr, Xj, v[p])
                    vj[s] = vj[s] + qd[j]
                for r in range(6):
                    self.put(tab.V + 6 * j + r, vj[r])
            else:
                vj = [tr.utab_get(tab.V + 6 * j + r) for r in range(6)] if p != -1 else None       # (a base joint's a does not need v)
            v[j] = vj
            xa = alg.matvec(tr, Xj, a[p]) if p != -1 else [Xj[r][5] * g for r in range(6)]
            aj = list(xa)
            if p != -1:
                aj = alg.vadd(aj, alg.mxS(tr, s, vj, qd[j]))
            if qdd is not None:
                aj[s] = aj[s] + qdd[j]
            a[j] = aj
            for r in range(6):
                self.put(tab.A + 6 * j + r, aj[r])
                if park_grad:
                    self.put(tab.XA + 6 * j + r, xa[r])
        tr.wave_sync()
        al = [self.lane_tab(tab.A, r) for r in range(6)]
        if first:
            self.own_from_table()
        fl = alg.vadd(self.sym_mv(al), self.own["fxv"])
        f = [None] * m
        c = [None] * m
        for j in range(m - 1, -1, -1):
            p, s = sub.parent[j], sub.S_ind[j]
            fj = [tr.bcast(fl[r], j) for r in range(6)]
            if f[j] is not None:
                fj = alg.vadd(fj, f[j])
            c[j] = fj[s] + qd[j] * sub.damping[j]
            if park_grad:
                for r in range(6):
                    self.put(tab.F + 6 * j + r, fj[r])
            if p != -1:
                f[p] = alg.mattvec_acc(tr, self.X_of(j, q, trig), fj, f[p] if f[p] is not None else alg.zeros6(tr))
        return c


WAVE_KINDS = ("id", "minv", "fd", "id_du", "fd_du")


def core_forward_dynamics_gradient_wave(sub, helped=False, helper_for=None, barriers=False, kind="fd_du"):
    """Forward-dynamics gradient (kind "fd_du") of ONE configuration on one wavefront, for the sub-forest `sub` (model.SubForest or
    a whole RobotSpec) -- or one of the other four algorithms from the same phases (the small-batch path of every kernel):
    "id" RNEA bias forces c (qdd from in.lane_qdd(): zero when the caller has none), "minv" the upper triangle of Minv (lane k:
    column k), "fd" qdd (lane k: qdd_k), "id_du" the RNEA gradient columns.

    Roles inside a block (wave_roles): a wave with slack can run the first RNEA pass of ANOTHER wave's group while that wave is busy
    with the articulated-inertia and Minv recursions (which do not need it): helper_for = that group's SubForest -- the pass writes
    v, a and the bias forces c into the other wave's table (accessor methods with suffix 2), then both meet at ONE block barrier;
    helped = True: this wave skips its first pass and takes c from its table after the barrier.  barriers = True: a wave that is
    neither still executes the block's barrier.  Inputs: in.lane_q/qd/u() = q, qd, u of joint (lane mod m); masks in.mask_k(j) = [lane mod m == j],
    in.mask_dq(j) = [lane == j], in.mask_dqd(j) = [lane == m + j].  Outputs put(r, value): row r of the lane's gradient column."""
    m = sub.n
    tr = Tracer()
    tab = WaveTable(m)
    serial = [0]

    def fresh(expr):
        serial[0] += 1
        return tr.inp("%s/*%d*/" % (expr, serial[0]))

    g = tr.inp("gravity")
    if helper_for is not None:
        # helper role first: the other group's first RNEA pass into the other wave's table, its bias forces behind it
        mk_h = tr.cse_mark()
        other = _Rnea(tr, helper_for, g, "2")
        c_other = other.run(None, True)
        for j in range(helper_for.n):
            other.put(other.tab.C + j, c_other[j])
        tr.barrier()
        tr.fence()
        tr.cse_release(mk_h)
    elif barriers and not helped:
        tr.barrier()
    rn = _Rnea(tr, sub, g)
    ql, qdl, sl, cl = rn.ql, rn.qdl, rn.sl, rn.cl
    ul = tr.inp("in.lane_u()") if kind in ("fd", "fd_du") else None
    I = alg.build_I(tr, sub)
    uniform_inputs, X_of, own = rn.uniform_inputs, rn.X_of, rn.own

    # ---- phase 1: articulated-inertia recursion (uniform) + the Minv recursions with lane = column k ------------------------------
    need_minv = kind in ("minv", "fd", "fd_du")
    mark = tr.cse_mark()
    Mcol = None
    if need_minv:
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

    if kind == "minv":
        for j in range(m):
            tr.out(j, Mc[j] * fresh("in.mask_row_le(%d)" % j))        # the upper triangle: rows j <= the lane's column, zeros below
        return tr

    # ---- phases 2 + 3: RNEA at qdd = 0 (bias forces c), qdd, RNEA at qdd; v, X a_parent and the accumulated f go to the table --------
    if kind in ("id", "id_du"):
        # inverse dynamics (and its gradient) at a GIVEN qdd: one pass; in.lane_qdd() is the lane's joint's qdd (zero when the caller
        # passed none: the kernels of the reference that are "optimized for qdd = 0" are the same arithmetic with fewer terms)
        qdd_in = tr.inp("in.lane_qdd()")
        c = rn.run([tr.bcast(qdd_in, j) for j in range(m)], True, park_grad=(kind == "id_du"))
        if kind == "id":
            for j in range(m):
                tr.out(j, c[j])
            return tr
        tr.fence()
        tr.cse_release(mark, keep=rn.own_refs())
    else:
        if helped:
            tr.barrier()                           # the helper wave has parked v and the bias forces c in this wave's table
            tr.fence()
            rn.own_from_table()
            c = [tr.utab_get(tab.C + j) for j in range(m)]
        else:
            c = rn.run(None, True)
        with tr.mixed_region():
            umc = [tr.bcast(ul, j) - c[j] for j in range(m)]
            qdd_lane = tr.dot([(Mcol[j], umc[j]) for j in range(m)])             # lane k: qdd_k = sum_j Minv[j][k] (u_j - c_j)
        qdd_lane = tr.cast(qdd_lane, 0) if tr.mixed else qdd_lane
        if kind == "fd":
            tr.out(0, qdd_lane)
            return tr
        tr.anchor(qdd_lane)
        tr.fence()
        tr.cse_release(mark, keep=rn.own_refs())  # (Minv columns are in LDS, the bias forces are folded into qdd: only the lane's own v, I v survive)
        mk = tr.cse_mark()
        qdd = [tr.bcast(qdd_lane, j) for j in range(m)]
        rn.run(qdd, False)
        tr.fence()
        tr.cse_release(mk, keep=rn.own_refs())

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
        Iv = [tr.bcast(own["Iv"][r], j) for r in range(6)]          # I_j v_j was formed by the lane that owns joint j
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

    if kind == "id_du":
        for r in range(m):
            tr.out(r, dc[r])
        return tr
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
