"""Build the traces ("cores") for each public algorithm variant.

A core is one straight-line function ``name<T, C>(in, out, gravity)`` over an input accessor
(``in.q(j)``, ``in.qd(j)``, ``in.u(j)``, ``in.qdd(j)``, ``in.Minv(i)``) and an output sink
(``out.put(i, value)``); kernels plug LDS-staged accessors in, the pointer-style ``_device`` API and
the host test harness plug plain-pointer accessors in.

Output index conventions (the reference's buffer layouts):
    c, qdd            i = joint
    Minv              i = n*col + row, column-major, upper triangle, lower half written as 0
    dc_du / df_du     i = n*col + row for d/dq (col < n), n*n + n*col + row for d/dqd
"""
from . import algorithms as alg
from .trace import Tracer


def _inputs(tr, spec, names, style):
    n = spec.n
    if style == "core":
        return {nm: [tr.inp("in.%s(%d)" % (nm, j)) for j in range(n)] for nm in names}
    return {nm: [tr.inp("s_%s[%d]" % (nm, j)) for j in range(n)] for nm in names}


def _setup(spec, names, style="core"):
    """style "core": accessor inputs, sin/cos traced from q.  style "inner": the reference's pointer
    API (s_q, s_qd, ...) with sin/cos read from the per-lane s_XImats = [sin q | cos q] table that
    load_update_XImats_helpers fills (the lane-private analogue of helpers/_topology_helpers.py:90-182)."""
    tr = Tracer()
    ins = _inputs(tr, spec, names, style)
    g = tr.inp("gravity")
    if style == "core":
        trig = alg.trig_from_q(tr, spec, ins["q"])
    else:
        trig = [(tr.inp("s_XImats[%d]" % j), tr.inp("s_XImats[%d]" % (spec.n + j))) if spec.uses_trig[j] else None
                for j in range(spec.n)]
    X = alg.build_X(tr, spec, ins["q"], trig)
    I = alg.build_I(tr, spec)
    return tr, ins, g, X, I


def core_inverse_dynamics(spec, use_qdd):
    tr, ins, g, X, I = _setup(spec, ["q", "qd"] + (["qdd"] if use_qdd else []))
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], ins.get("qdd"), g)
    for j in range(spec.n):
        tr.out(j, c[j])
    return tr


def core_inverse_dynamics_vaf(spec, use_qdd):
    tr, ins, g, X, I = _setup(spec, ["q", "qd"] + (["qdd"] if use_qdd else []))
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], ins.get("qdd"), g)
    n = spec.n
    for j in range(n):
        for r in range(6):
            tr.out(6 * j + r, v[j][r])
    for j in range(n):
        for r in range(6):
            tr.out(6 * n + 6 * j + r, a[j][r])
    for j in range(n):
        for r in range(6):
            tr.out(12 * n + 6 * j + r, f[j][r])
    return tr


def _out_minv(tr, spec, Minv):
    n = spec.n
    for col in range(n):
        for row in range(n):
            tr.out(n * col + row, Minv[row][col] if row <= col else tr.zero())


def core_direct_minv(spec):
    tr, ins, g, X, I = _setup(spec, ["q"])
    Minv = alg.direct_minv(tr, spec, X, I)
    _out_minv(tr, spec, Minv)
    return tr


def core_forward_dynamics(spec):
    tr, ins, g, X, I = _setup(spec, ["q", "qd", "u"])
    Minv = alg.direct_minv(tr, spec, X, I)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], None, g)
    qdd = alg.fd_finish(tr, spec, Minv, ins["u"], c)
    for j in range(spec.n):
        tr.out(j, qdd[j])
    return tr


def _out_grad(tr, spec, G, cols=None):
    """G[row][col]: pair (d/dq, d/dqd).  cols=None: all 2n^2 outputs at their global index.  cols=[c0..c1]
    (contiguous): only those columns of both halves, at LOCAL indices 0..2*n*len(cols)-1 (d/dq columns first, then
    d/dqd); the kernel sink maps the two local runs back to n*c0 and n*n + n*c0 (column-split kernels; dead-code
    elimination drops everything the other columns needed)."""
    n = spec.n
    if cols is None:
        for col in range(n):
            for row in range(n):
                tr.out(n * col + row, G[row][col].lo)
        for col in range(n):
            for row in range(n):
                tr.out(n * n + n * col + row, G[row][col].hi)
        return
    i = 0                                   # (cols need not be contiguous: the column-set sink maps every column back)
    lo_cols, hi_cols = cols if isinstance(cols, tuple) else (cols, cols)      # a tuple: different column sets for d/dq and d/dqd
    for half, group in (("lo", lo_cols), ("hi", hi_cols)):
        for col in group:
            for row in range(n):
                tr.out(i, getattr(G[row][col], half))
                i += 1


def core_inverse_dynamics_gradient(spec, use_qdd, cols=None):      # cols: None | [columns] | ([d/dq columns], [d/dqd columns])
    tr, ins, g, X, I = _setup(spec, ["q", "qd"] + (["qdd"] if use_qdd else []))
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], ins.get("qdd"), g)
    dc = alg.rnea_grad(tr, spec, X, I, ins["qd"], v, a, f, g)
    _out_grad(tr, spec, dc, cols)
    return tr


def core_forward_dynamics_gradient(spec, use_qdd_minv, cols=None):
    n = spec.n
    if use_qdd_minv:
        tr, ins, g, X, I = _setup(spec, ["q", "qd", "qdd"])
        Min = [tr.inp("in.Minv(%d)" % i) for i in range(n * n)]
        Minv = [[Min[n * c + r] if r <= c else None for c in range(n)] for r in range(n)]
        qdd = ins["qdd"]
    else:
        tr, ins, g, X, I = _setup(spec, ["q", "qd", "u"])
        Minv = alg.direct_minv(tr, spec, X, I)
        c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], None, g)
        qdd = alg.fd_finish(tr, spec, Minv, ins["u"], c)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], qdd, g)
    dc = alg.rnea_grad(tr, spec, X, I, ins["qd"], v, a, f, g)
    out = alg.fd_grad_finish(tr, spec, Minv, dc, sorted(set(cols[0]) | set(cols[1])) if isinstance(cols, tuple) else cols)
    _out_grad(tr, spec, out, cols)
    return tr


class CoopSlots:
    """Exchange-region slots of the tile-cooperative forward-dynamics-gradient kernels: the structurally non-zero upper
    triangle of Minv, then the bias torques c and the accelerations qdd (two shares when two producer waves split the forward
    pass of the Minv recursion).  Value `slot` of lane l lives at slot*64 + l of the block's LDS region."""

    def __init__(self, spec):
        n = spec.n
        nz = alg.minv_zero_pattern(spec)
        self.minv = {}
        for r in range(n):
            for k in range(r, n):
                if nz[r][k]:
                    self.minv[(r, k)] = len(self.minv)
        base = len(self.minv)
        self.c = [base + j for j in range(n)]
        self.qdd = [base + n + j for j in range(n)]            # qdd = Minv (u - c), or the first producer's share of it
        self.qdd2 = [base + 2 * n + j for j in range(n)]       # the second producer's share (ksplit is not None)
        self.count = base + 3 * n
        # Wave-private use of the region (recomputing schedule): the accumulated force f_j of a gradient column's joint, parked by the
        # wave that owns column j and re-read when it reaches the column of j's parent (core_gradient_recompute: f table).  No
        # barrier involved -- a wave reads only what it wrote itself, and the DS operations of a wave execute in order.
        self.f = {}
        self.f_table = True              # (False: every column walks its joint's whole subtree again -- kept for the A/B test)
        for j in range(n):
            if spec.parent[j] != -1:
                self.f[j] = self.count
                self.count += 6
        # Two producer waves (large robots): both run the backward pass of the Minv recursion, then "producer" finishes the
        # columns k < ksplit of the forward pass and "producer2" the columns k >= ksplit (the forward pass is independent per
        # column); each folds its entries into its own share of qdd.  None: one producer does everything.
        self.ksplit = None
        # Consumer waves idle while the producers run the Minv recursion.  The d/dqd half of a gradient column needs neither qdd
        # nor Minv before its final product, so a consumer computes it for some of its columns BEFORE the first barrier and keeps
        # the n values per column in registers: hoist_budget[role] = arithmetic instructions of idle time to fill,
        # hoist_cost[col] = what the d/dqd recursion of that column costs (None: no hoisting).
        self.hoist_budget = None
        self.hoist_cost = None
        self.hoist_max_columns = 4       # n parked values per column stay in registers across the barriers

    lean = False            # see enable_lean()
    lean_umc = False
    itab = None

    def enable_lean(self, spec, umc=False):
        """Register-lean cores for blocks whose waves PAIR UP on the SIMDs (8 waves per tile, <= 256 registers each; the 4-wave
        kernel's waves own a SIMD and 464 registers).  What changes in the cores (core_gradient_recompute):
          * sin q, cos q, qd (q of prismatic joints) live in a block-shared input table of the exchange region -- every wave writes the
            same values before it reads any (DS operations of a wave execute in order; the other waves' writes are bit-identical) --
            and are re-read where a joint is touched instead of staying in ~4n registers; qdd is re-read from the producers' slots;
          * X_j is rebuilt for the way back up the tree instead of staying alive across the child's subtree (12 registers per level);
          * the chain root -> column joint is traced joint by joint (X_j, a_j, v_j), the subtree's v, a are forgotten after the walk
            that accumulates the column joint's force;
          * no f table (its 41 KB are the staging regions of the four extra waves).
        A wave's work is a list of HALF columns [(column, 0 = d/dq | 1 = d/dqd)]: the two halves of a column share nothing but v and
        X, so they can run on different waves, each carrying half of the gradient state."""
        n = spec.n
        self.lean = True
        self.f = {}
        self.f_table = False
        base = len(self.minv)
        self.qdd2 = None
        self.lean_umc = bool(umc)
        if umc:
            # the waves that compute the bias torques publish u - c (the only form c is used in: qdd = Minv (u - c)) and read u
            # themselves: no u in the input table, and the n words live BELOW the region next to U and 1/D (the staging regions,
            # idle until the first flush): 2 n slots = 15 KB less LDS for Atlas-30, which the 34-word staging rows of the
            # sector-aligned flush need
            self.c = [-(7 * n + 1 + j) for j in range(n)]
            self.qdd = [base + j for j in range(n)]
            self.count = base + n
            self.itab = {"s": [self.count + j for j in range(n)], "c": [self.count + n + j for j in range(n)],
                         "qd": [self.count + 2 * n + j for j in range(n)]}
            self.count += 3 * n
        else:
            self.count = base + 2 * n               # c, qdd (one producer of qdd per row: no second share)
            self.itab = {"s": [self.count + j for j in range(n)], "c": [self.count + n + j for j in range(n)],
                         "qd": [self.count + 2 * n + j for j in range(n)], "u": [self.count + 3 * n + j for j in range(n)]}
            self.count += 4 * n
        if any(not t for t in spec.uses_trig):
            self.itab["q"] = [self.count + j for j in range(n)]
            self.count += n

    def hoisted_columns(self, role, cols):
        """The columns of `cols` whose d/dqd recursion this role runs ahead of the barriers: cheapest first while the budget lasts."""
        if not self.hoist_budget or role not in self.hoist_budget or not cols:
            return []
        left, picked = self.hoist_budget[role], []
        for c in sorted(cols, key=lambda c: self.hoist_cost[c]):
            if self.hoist_cost[c] > left or len(picked) >= self.hoist_max_columns:
                break
            picked.append(c)
            left -= self.hoist_cost[c]
        return picked

    def entry(self, tr, r, k):
        """Minv_sym[r][k] as a fresh exchange read (None: structural zero)."""
        slot = self.minv.get((r, k) if r <= k else (k, r))
        return tr.xch_get(slot) if slot is not None else None


COOP_ROLES = ("producer", "producer2", "consumer_c", "consumer")


class AlignedPieces:
    """Output stream of a lean core whose half-columns arrive in runs of NEIGHBOURS of the output row (lean_plan(order="runs")): the
    n values of a half-column are not flushed as they are -- 120 bytes at an 8-byte aligned offset: every flush leaves two partly
    written 32-byte sectors per configuration, which the memory side writes as whole sectors (WRITE_SIZE 1.23x the output, and the
    pure store stream of K = 65536 takes 187 us against 159 us in aligned pieces: profiles/r04/ubench_store_pieces.txt) -- but cut at
    the sector boundaries of the row: the values past the last boundary (below the first one, when the run is walked downwards)
    stay in registers (at most 7) and leave with the next half-column, which continues the same run.  A piece is at most 32
    values; only the two ends of a run write partial sectors.
    The row itself starts on a sector boundary when 2 n^2 is a multiple of 8 (n even); otherwise no cutting.

    Emits, per piece: tr.out("piece:<len>:<pos>", value) for its values, then tr.out("flush:<len>:<row offset>", 0)."""

    SECTOR = 8          # values of the storage type (float) per 32-byte sector
    MAX_PIECE = 32      # values per piece (grid_out_pieces: staging pitch 34)

    def __init__(self, tr, n, row, bases):
        self.tr, self.n, self.bases = tr, n, list(bases)
        self.unit = self.SECTOR if row % self.SECTOR == 0 else 1
        self.k = 0
        self.lo, self.vals = None, []       # pending values: row offsets lo .. lo + len(vals)
        self.pieces = []                    # (row offset, length) of every emitted piece, in order

    def _emit(self, lo, hi):
        """Write the pending values of row offsets [lo, hi) -- one end of the pending interval -- in pieces of at most MAX_PIECE values
        whose inner boundaries are sector boundaries."""
        first = lo - self.lo
        vals = self.vals[first:first + hi - lo]
        at = lo
        while at < hi:
            stop = min(hi, ((at + self.MAX_PIECE) // self.unit) * self.unit if hi - at > self.MAX_PIECE else hi)
            length = stop - at
            for pos in range(length):
                self.tr.out("piece:%d:%d" % (length, pos), vals[at - lo + pos])
            self.tr.out("flush:%d:%d" % (length, at), 0.0)
            self.pieces.append((at, length))
            at = stop
        if lo == self.lo:
            self.vals, self.lo = self.vals[hi - lo:], hi
        else:
            assert hi == self.lo + len(self.vals)
            self.vals = self.vals[:first]

    def push(self, base, values):
        assert base == self.bases[self.k] and len(values) == self.n
        self.k += 1
        if not self.vals:
            self.lo, self.vals = base, list(values)
        elif self.lo + len(self.vals) == base:          # continues the run upwards
            self.vals += list(values)
        else:                                           # ... or downwards
            assert base + self.n == self.lo
            self.lo, self.vals = base, list(values) + self.vals
        lo, hi = self.lo, self.lo + len(self.vals)
        nxt = self.bases[self.k] if self.k < len(self.bases) else None
        if nxt is not None and nxt == hi:               # the next half-column continues upwards: keep what lies past the last boundary
            cut = (hi // self.unit) * self.unit
            if cut > lo:
                self._emit(lo, cut)
        elif nxt is not None and nxt + self.n == lo:    # ... downwards: keep what lies below the first boundary
            cut = -((-lo) // self.unit) * self.unit
            if cut < hi:
                self._emit(cut, hi)
        else:
            self._emit(lo, hi)


class LeanRole:
    """What ONE wave of a register-lean tile-cooperative block (CoopSlots.enable_lean, 8 waves per tile) does before its gradient
    half-columns:

        joints      phase 0: the joints whose sin q, cos q, qd, u this wave writes to the block's input table        -> barrier B0
        minv_bwd    phase 1: joints (whole base-rooted trees) whose BACKWARD pass of the Minv recursion this wave runs, publishing
                    U, 1/D and the backward-pass entries (alg.minv_backward_lean) -- once per tree, shared by every wave
        c_roots     phase 1: base joints of the trees whose bias torques c = RNEA(q, qd, 0) this wave computes (depth first: only one
                    root-to-leaf path of v, a, f alive) and publishes
        hoist       phase 1: columns whose d/dqd recursion runs here, ahead of the barriers (n parked values per column)
                                                                                                                      -> barrier B1
        minv_cols   phase 2: the columns of Minv whose FORWARD pass this wave runs (alg.minv_forward_lean: independent per column,
                    so all waves share it)                                                                            -> barrier B2
        qdd_rows    phase 3: rows r of qdd = Minv_sym (u - c) this wave computes from the published Minv and c        -> barrier B3
    Every wave executes the same four barriers."""

    out_qdd = False         # (forward-dynamics kernel on the same block, lean_plan_fd: this wave writes qdd to the output after B3)
    out_minv = False        # (Minv kernel on the same block, lean_plan_minv: this wave writes its forward-pass columns of Minv after B1)

    def __init__(self, name="consumer", joints=(), minv_bwd=(), minv_cols=(), c_roots=(), hoist=(), qdd_rows=(), minv_bwd_cols=None):
        self.name, self.joints, self.minv_bwd, self.minv_cols = name, list(joints), list(minv_bwd), sorted(minv_cols)
        self.minv_bwd_cols = None if minv_bwd_cols is None else sorted(minv_bwd_cols)     # (only these columns of the backward pass)
        self.c_roots, self.hoist, self.qdd_rows = list(c_roots), list(hoist), list(qdd_rows)

    def __repr__(self):
        return "LeanRole(%s: table %s, Minv backward pass of %s%s, c of trees %s, parked %s, Minv forward columns %s, qdd rows %s)" % (
            self.name, self.joints, self.minv_bwd, "" if self.minv_bwd_cols is None else " columns %s" % self.minv_bwd_cols,
            self.c_roots, self.hoist, self.minv_cols, self.qdd_rows)


LEAN_BARRIERS = 4
FWD_BIAS = (0.0,)       # (planner: extra cost charged to wave w's forward-pass share; one entry = the same for every wave)
LEAN_WAVES = 8
LEAN_YOUNGER_SPEED = 1.0        # weight of the waves dispatched second when the gradient half-columns are dealt out.  Issue-bound code
                                # gives the younger wave of a SIMD 0.57-0.67 of the older one's pace (profiles/r03/two_waves_per_simd.md);
                                # these cores wait on LDS and on dependent chains (6.3 cycles per instruction alone), and equal shares
                                # measured best: K = 16384 57.5 us (0.6), 55.3 (0.8), 54.2 (1.0), 56.3 (1.25) -- profiles/r04/lean_probe_2.txt
LEAN_FLUSH_SLOTS = 150          # issue slots one half-column flush (n LDS writes, a wave sync, the paired reads and stores) costs


def lean_arith(tr, start=1, stop=None):
    """Arithmetic instructions + LDS reads of the live part of a trace between two node positions."""
    live = tr.live_nodes()
    stop = len(tr.nodes) if stop is None else stop
    return sum(1 for k in range(start, stop) if live[k] and tr.nodes[k][0] in ("fma", "mul", "add", "rcp", "in", "pkfma", "pkmul", "pkadd"))


def lean_barriers(tr):
    return [pos for (dst, _), pos in zip(tr.outputs, tr.out_pos) if dst == "barrier"]


def lean_plan(spec, waves=LEAN_WAVES, max_parked=3, younger_speed=None, keep_x_below=0, columns_from_chain=False, order="runs",
              products_per_half=False, separate_halves=False, umc=True, aligned_flush=True, chain_f=True, pair_products=False):
    """Who does what in a register-lean block of `waves` wavefronts (two per SIMD): returns (slots, [(LeanRole, [(column, half)])]).

    Phase 0 (input table): joints dealt round-robin.  Phase 1: the BACKWARD pass of the Minv recursion once per base-rooted tree -- the
    largest tree on wave 0, the others on wave 1, which also computes their bias torques; wave 2 computes the largest tree's bias
    torques.  Waves 0-3 are dispatched first (the OLDER wave of each SIMD, which keeps the lone-wave pace) and take these roles;
    the waves that are free run d/dqd recursions ahead of the barrier (parked, at most `max_parked` columns of n values each).
    Phase 2: the FORWARD pass, independent per column, divided over all waves in contiguous runs of balanced cost.  Phase 3: qdd rows
    round-robin.  Gradient half-columns: longest-processing-time placement on traced per-item costs, a younger wave's items
    weighted by 1 / younger_speed (default LEAN_YOUNGER_SPEED).

    order="runs": every wave takes a CONTIGUOUS run of gradient columns, both halves of each (the d/dq and the d/dqd column share
    the walk over the joints), instead of the scattered longest-processing-time sets: the half-columns a wave flushes one after
    the other are then neighbours in the configuration's output row, so the partly written 64-byte segments at the ends of a run
    of n values meet their other part in L2 within one column's time instead of reaching HBM twice (WRITE_SIZE 1.3x the output at
    K = 65536, where the output no longer fits the Infinity Cache).  The runs and which role takes which run: a dynamic programme
    over the cut points for every assignment of the roles' idle times, on traced per-column costs."""
    from .model import base_trees
    n = spec.n
    slots = CoopSlots(spec)
    slots.enable_lean(spec, umc=umc)
    slots.keep_x_below = keep_x_below
    slots.columns_from_chain = bool(columns_from_chain)
    slots.products_per_half = bool(products_per_half)
    slots.separate_halves = bool(separate_halves)
    slots.aligned_flush = bool(aligned_flush)
    slots.chain_f = bool(chain_f)
    slots.pair_products = bool(pair_products)
    trees = sorted(base_trees(spec), key=lambda t: -t[1])
    big = list(range(trees[0][0], trees[0][0] + trees[0][1]))
    rest = [j for (f, m) in trees[1:] for j in range(f, f + m)]
    rest_roots = [f for (f, m) in trees[1:]]

    def probe(role, items):
        return core_gradient_recompute(spec, "fd", cols=items, coop=(role, slots))
    # the largest tree's backward pass on two waves: both run the articulated-inertia chain (U, 1/D), each the F recursions of its
    # share of the columns -- the cut that balances the two (the early columns walk further up the tree)
    def bwd_cost(cols_):
        tr = probe(LeanRole("minv_backward", minv_bwd=big, minv_bwd_cols=cols_), [(n - 1, 1)])
        b = lean_barriers(tr)
        return lean_arith(tr, b[0], b[1])
    cut_b = min(range(big[0] + 1, big[-1] + 1), key=lambda c_: max(bwd_cost([k for k in big if k < c_]), bwd_cost([k for k in big if k >= c_])))
    if columns_from_chain:
        # only the articulated-inertia chain (U, 1/D) is serial: one wave per group of trees runs it, everything per column follows B1
        roles = [LeanRole("inertia_chain", minv_bwd=big, minv_bwd_cols=[]),
                 LeanRole("inertia_chain+c", minv_bwd=rest, minv_bwd_cols=[], c_roots=rest_roots) if rest else LeanRole("consumer"),
                 LeanRole("c", c_roots=[big[0]])]
    else:
        roles = [LeanRole("minv_backward_a", minv_bwd=big, minv_bwd_cols=[k for k in big if k < cut_b]),
                 LeanRole("minv_backward_b", minv_bwd=big, minv_bwd_cols=[k for k in big if k >= cut_b]),
                 LeanRole("minv_backward+c", minv_bwd=rest, c_roots=rest_roots) if rest else LeanRole("consumer"),
                 LeanRole("c", c_roots=[big[0]])]
    roles += [LeanRole("consumer") for _ in range(waves - len(roles))]
    # forward pass: a wave that finishes columns a..b-1 walks every joint j < b of their trees (fixed cost per joint: U, 1/D, X_j, X_j^T U)
    # and pays per (joint, column >= joint) pair; contiguous runs share the joints.  Minimise the largest run (dynamic programme).
    tree_of = {}
    for (f, m) in base_trees(spec):
        for j in range(f, f + m):
            tree_of[j] = f
    PER_JOINT, PER_PAIR = (70.0, 30.0) if columns_from_chain else (45.0, 16.0)

    def run_cost(a, b):
        cost_, seen = 0.0, set()
        for k in range(a, b):
            for j in range(tree_of[k], k + 1):
                if j not in seen:
                    seen.add(j)
                    cost_ += PER_JOINT
                cost_ += PER_PAIR
        return cost_
    INF = float("inf")
    best = [[INF] * (n + 1) for _ in range(waves + 1)]
    cut = [[0] * (n + 1) for _ in range(waves + 1)]
    best[0][0] = 0.0
    for w in range(1, waves + 1):
        for e in range(w, n + 1):
            for b_ in range(w - 1, e):
                if best[w - 1][b_] < INF:
                    c_ = max(best[w - 1][b_], run_cost(b_, e) + FWD_BIAS[min(w - 1, len(FWD_BIAS) - 1)])
                    if c_ < best[w][e]:
                        best[w][e], cut[w][e] = c_, b_
    e = n
    for w in range(waves, 0, -1):
        roles[w - 1].minv_cols = list(range(cut[w][e], e))
        e = cut[w][e]
    for w, role in enumerate(roles):
        role.joints = [j for j in range(n) if j % waves == w]
        role.qdd_rows = [j for j in range(n) if j % waves == w]
    lean_partial_layout(spec, slots, roles)

    def phase_costs(role):
        tr = probe(role, [(n - 1, 1)])
        b = lean_barriers(tr)
        return lean_arith(tr, b[0], b[1]), lean_arith(tr, b[1], b[2])
    ph = [phase_costs(r) for r in roles]
    t_b1 = max(p1 for (p1, _) in ph)
    # per-item costs: the recursion (what parking moves ahead of the barrier) and the whole item
    cost, rec = {}, {}
    for c_ in range(n):
        for h in (0, 1):
            tr = probe(LeanRole("consumer"), [(c_, h)])
            b = lean_barriers(tr)
            cost[(c_, h)] = lean_arith(tr, b[-1]) + LEAN_FLUSH_SLOTS
        trp = probe(LeanRole("consumer", hoist=[c_]), [(c_, 1)])
        b = lean_barriers(trp)
        rec[c_] = lean_arith(trp, b[0], b[1])
    speed = [1.0 if w < waves // 2 else (LEAN_YOUNGER_SPEED if younger_speed is None else younger_speed) for w in range(waves)]
    if order == "runs":
        top_extra = _chained_costs(spec, cost, lambda items: (lambda tr: lean_arith(tr, lean_barriers(tr)[-1]))(probe(LeanRole("consumer"), items))) if chain_f else None
        return slots, _lean_plan_runs(spec, slots, roles, cost, rec, [t_b1 - p1 for (p1, _) in ph], max_parked, speed, ph, t_b1,
                                      prefix_parking=aligned_flush, top_extra=top_extra)
    assert order == "lpt", order
    load = [0.0] * waves
    items = [[] for _ in range(waves)]
    parked = [[] for _ in range(waves)]
    slack = [t_b1 - ph[w][0] for w in range(waves)]
    for it in sorted(cost, key=lambda it: -cost[it]):
        if cost[it] <= LEAN_FLUSH_SLOTS:
            continue                                   # structurally zero half-column: placed last, wherever it is cheapest

        def finish(w):
            c_, h = it
            gain = rec[c_] if (h == 1 and len(parked[w]) < max_parked and rec[c_] <= slack[w]) else 0
            return (load[w] + cost[it] - gain) / speed[w], gain
        w = min(range(waves), key=lambda w: finish(w)[0])
        gain = finish(w)[1]
        if gain:
            parked[w].append(it[0])
            slack[w] -= gain
        load[w] += cost[it] - gain
        items[w].append(it)
    for it in sorted(cost):
        if cost[it] <= LEAN_FLUSH_SLOTS:
            w = min(range(waves), key=lambda w: (load[w] + cost[it]) / speed[w])
            load[w] += cost[it]
            items[w].append(it)
    for w, role in enumerate(roles):
        role.hoist = sorted(parked[w])
    plan = [(roles[w], sorted(items[w])) for w in range(waves)]
    slots.lean_model = dict(phase1=[p1 for (p1, _) in ph], phase2=[p2 for (_, p2) in ph], t_b1=t_b1, post=[load[w] / speed[w] for w in range(waves)])
    return slots, plan


def lean_partial_layout(spec, slots, roles):
    """Mixed arithmetic on the lean block: qdd = Minv (u - c) must come from the UNROUNDED Minv (from the float copy in LDS the
    gradient loses what the double recursion bought: profiles/r04/mixed_lean_report.txt).  Every wave therefore folds the final
    entries of its forward-pass columns into partial sums of qdd in double while they are still in registers, and publishes them
    as float pairs (hi, lo) after the forward pass -- in the staging regions, where U, 1/D and u - c are dead by then.  Lays out:
    role.index; slots.partials[w] = (first pair, rows the wave contributes to); slots.root_lo[(root, k)] = word for the low part of
    a base joint's row of Minv (final in the backward pass already, published there as hi + lo)."""
    n = spec.n
    slots.partials, off = {}, 0
    for w, role in enumerate(roles):
        role.index = w
        rows = sorted(set(k for k in role.minv_cols) | set(r for k in role.minv_cols for r in range(n) if (r, k) in slots.minv and r <= k))
        slots.partials[w] = (off, rows)
        off += len(rows)
    slots.partial_pairs = off
    slots.root_lo, at = {}, 8 * n
    for j in range(n):
        if spec.parent[j] == -1:
            for k in spec.subtree[j]:
                at += 1
                slots.root_lo[(j, k)] = -at
    slots.scratch_words = max(2 * off, at)          # words per lane needed below the exchange region


def _chained_costs(spec, cost, post_arith):
    """chain_f: replaces cost[(c, 0)] by what the d/dq half of column c costs right behind column c + 1 in the same wave (the
    accumulated force of c + 1 is reused when c + 1 is a child of c) and returns top_extra[c] = what it costs more on its own."""
    n = spec.n
    top_extra = [0.0] * n
    for c_ in range(n - 1):
        if spec.parent[c_ + 1] == c_ and spec.parent[c_] != -1:
            behind = post_arith([(c_ + 1, 0), (c_, 0)]) - post_arith([(c_ + 1, 0)]) + LEAN_FLUSH_SLOTS
            top_extra[c_] = max(0.0, cost[(c_, 0)] - behind)
            cost[(c_, 0)] = min(cost[(c_, 0)], behind)
    return top_extra


def lean_plan_fd(spec, waves=LEAN_WAVES, **options):
    """The register-lean block as a FORWARD-DYNAMICS kernel (qdd = Minv (u - c) only): the prefix of lean_plan -- input table, Minv
    recursion shared through LDS, bias torques, qdd rows -- without gradient columns and without parked recursions; wave 0 writes
    the n accelerations of the tile's configurations after B3.  Same arithmetic as the prefix of the gradient kernel (and, to three
    digits, the accuracy of the lane-per-configuration forward-dynamics core)."""
    slots, plan = lean_plan(spec, waves, **options)
    out = []
    for w, (role, _) in enumerate(plan):
        role.hoist = []
        role.out_qdd = (w == 0)
        out.append((role, []))
    return slots, out


def lean_plan_minv(spec, waves=LEAN_WAVES, **options):
    """The register-lean block as a DIRECT-Minv kernel: input table (sin, cos only: the input rows may hold nothing but q), backward pass
    of the recursion once per base-rooted tree, forward pass over all eight waves -- and every wave writes the columns it finishes,
    upper triangle, from registers (the rows of base joints, final since the backward pass, from LDS).  No bias torques, no qdd."""
    slots, plan = lean_plan(spec, waves, **options)
    out = []
    for role, _ in plan:
        role.hoist, role.c_roots, role.qdd_rows = [], [], []
        role.out_minv = True
        out.append((role, []))
    slots.table_q_only = True
    return slots, out


def lean_plan_id(spec, use_qdd=False, waves=LEAN_WAVES, chain_f=False):
    """Who does what in a register-lean block of the INVERSE-dynamics gradient (dc_du at (q, qd[, qdd]); 8 waves per tile, two per
    SIMD): every wave writes its share of the block's input table, one barrier, then its gradient half-columns -- one contiguous
    run of d/dq columns and one of d/dqd columns each, cut by the same dynamic programme as the forward-dynamics kernel's (no Minv,
    no bias torques, nothing to park).  Returns (slots, [(LeanRole, [(column, half)])]).

    chain_f (the accumulated force of the column finished last reused by its parent's column, as in lean_plan) is OFF here: it
    takes 12.7 % of the instructions away (54.6 k -> 47.7 k per Atlas-30 tile) and K = 64 from 23.0 to 20.4 us, but hipcc then needs
    more than 256 registers for the wave that holds the two heaviest d/dqd columns (104 B of scratch; 122 registers and none
    without) and the full chip gets slower: K = 16384 32.4 -> 33.4 us, K = 65536 206 -> 217 us (profiles/r04/lean_chain_f.txt)."""
    n = spec.n
    slots = CoopSlots(spec)
    slots.minv = {}                      # (no Minv in this kernel: the region holds the input table and, with use_qdd, qdd)
    slots.enable_lean(spec, umc=True)
    slots.keep_x_below = 0
    slots.columns_from_chain = False
    slots.products_per_half = False
    slots.separate_halves = False
    slots.aligned_flush = True
    slots.chain_f = bool(chain_f)
    roles = [LeanRole("columns", joints=[j for j in range(n) if j % waves == w]) for w in range(waves)]

    def post_arith(items):
        tr = core_gradient_recompute(spec, "id", use_qdd=use_qdd, cols=items, coop=(LeanRole("columns"), slots))
        return lean_arith(tr, lean_barriers(tr)[-1])
    cost = {(c_, h): post_arith([(c_, h)]) + LEAN_FLUSH_SLOTS for c_ in range(n) for h in (0, 1)}
    top_extra = _chained_costs(spec, cost, post_arith) if chain_f else None
    plan = _lean_plan_runs(spec, slots, roles, cost, [0] * n, [0] * waves, 0, [1.0] * waves, [(0, 0)] * waves, 0, top_extra=top_extra)
    return slots, plan


def _lean_plan_runs(spec, slots, roles, cost, rec, slack, max_parked, speed, ph, t_b1, prefix_parking=False, top_extra=None):
    """lean_plan(order="runs"): every wave takes ONE contiguous run of d/dq columns and ONE contiguous run of d/dqd columns.  The
    waves are taken in the order of their idle time before B1 (the busiest first); wave i gets the i-th run of the d/dq block
    counted from column 0 and the i-th run of the d/dqd block counted from column n-1 -- the heavy columns are the early ones in
    both blocks, so a heavy d/dq run meets a light d/dqd run, and the idle waves get the d/dqd columns worth parking.  The cuts of
    both blocks by one dynamic programme on the traced per-item costs minus what parking moves ahead of the barrier.  Returns
    the plan; sets slots.lean_model."""
    n, waves = spec.n, len(roles)
    pq, pd = [0.0], [0.0]
    for c_ in range(n):
        pq.append(pq[-1] + cost[(c_, 0)])
        pd.append(pd[-1] + cost[(n - 1 - c_, 1)])          # (position k of the d/dqd sequence is column n-1-k)
    # chain_f: cost[(c, 0)] is what column c costs BEHIND column c + 1 in the same wave; the highest column of a run pays top_extra[c] more
    top_of = (lambda a, e: top_extra[e - 1] if (top_extra is not None and e > a) else 0.0)
    gains = {}

    def parked_of(s, a, b_):                                # d/dqd positions [a, b_) = columns n-b_ .. n-1-a
        if (s, a, b_) not in gains:
            g, chosen = 0, []
            if prefix_parking:      # (the first columns of the run, in the order they are processed: AlignedPieces wants neighbours)
                for c_ in range(n - b_, n - a):
                    if len(chosen) >= max_parked or not (0 < rec[c_] <= s - g):
                        break
                    g += rec[c_]
                    chosen.append(c_)
            else:
                for c_ in sorted(range(n - b_, n - a), key=lambda c_: -rec[c_]):
                    if len(chosen) < max_parked and 0 < rec[c_] <= s - g:
                        g += rec[c_]
                        chosen.append(c_)
            gains[(s, a, b_)] = (g, chosen)
        return gains[(s, a, b_)]
    INF = float("inf")
    ws = sorted(range(waves), key=lambda w: (slack[w], -speed[w], w))
    best = {(0, 0): (0.0, None)}
    layers = [best]
    for i in range(1, waves + 1):
        w = ws[i - 1]
        dload = {}
        nxt = {}
        last = i == waves
        for (aq, ad), (top, _) in layers[-1].items():
            for eq in ((n,) if last else range(aq, n + 1)):
                lq = pq[eq] - pq[aq] + top_of(aq, eq)
                if lq / speed[w] >= (nxt.get((n, n), (INF,))[0] if last else INF):
                    break
                for ed in ((n,) if last else range(ad, n + 1)):
                    key = (ad, ed)
                    if key not in dload:
                        dload[key] = pd[ed] - pd[ad] - parked_of(slack[w], ad, ed)[0]
                    c_ = max(top, (lq + dload[key]) / speed[w])
                    if c_ < nxt.get((eq, ed), (INF,))[0]:
                        nxt[(eq, ed)] = (c_, (aq, ad))
        # keep the non-dominated states only (further in both blocks at no higher load)
        layers.append(nxt)
    state, runs = (n, n), []
    for i in range(waves, 0, -1):
        prev = layers[i][state][1]
        runs.append((prev[0], state[0], prev[1], state[1]))
        state = prev
    runs = runs[::-1]
    plan_of = {w: run for w, run in zip(ws, runs)}
    plan, post = [], []
    for w, role in enumerate(roles):
        aq, eq, ad, ed = plan_of[w]
        role.hoist = sorted(parked_of(slack[w], ad, ed)[1])
        plan.append((role, sorted([(c_, 0) for c_ in range(aq, eq)] + [(c_, 1) for c_ in range(n - ed, n - ad)])))
        post.append((pq[eq] - pq[aq] + top_of(aq, eq) + pd[ed] - pd[ad] - parked_of(slack[w], ad, ed)[0]) / speed[w])
    slots.lean_model = dict(phase1=[p1 for (p1, _) in ph], phase2=[p2 for (_, p2) in ph], t_b1=t_b1, post=post,
                            runs=[(plan_of[w][0], plan_of[w][1], n - plan_of[w][3], n - plan_of[w][2]) for w in range(waves)])
    return plan


def _coop_prologue(tr, spec, slots, role, X, I, qd, u, g, demand_order=True, pre_barrier=None):
    """Phases 1 and 2 of a tile-cooperative core; returns qdd (read from the exchange region by every wave).

    producer:  backward pass of the Minv recursion | barrier | c from the exchange region; forward pass: every entry of Minv is
               published as it becomes final and folded into qdd = Minv (u - c) right there (in double when the build is
               mixed-precision, so the cond(M)-amplified product sees the unrounded Minv) -> qdd to the exchange region | barrier
    producer2: (slots.ksplit set) the same backward pass, the columns k >= ksplit of the forward pass and its share of qdd
    consumers: RNEA at qdd = 0 while the producer is busy; one of them publishes c | barrier | barrier"""
    n = spec.n
    two = slots.ksplit is not None
    assert two or role != "producer2"
    if role in ("producer", "producer2"):
        # Minv entries are published -- and folded into qdd = Minv_sym (u - c) -- the moment they are final, so they never all
        # live at once (465 values for Atlas-30); c is needed from the forward pass on, so the first barrier sits between the
        # two passes of the recursion (the consumers have long finished RNEA by then).  With two producers each finishes,
        # publishes and folds only its own columns; the other columns' forward-pass arithmetic is dead code in its trace.
        state = {}
        mine = (lambda k: True) if not two else ((lambda k: k < slots.ksplit) if role == "producer" else (lambda k: k >= slots.ksplit))
        out_slots = slots.qdd if role == "producer" else slots.qdd2

        def between(carried):
            if demand_order:             # demand-order emission: the backward pass must be issued BEFORE the wave waits
                for val in carried:      # (creation-order emission places the barrier where it is traced)
                    tr.anchor(val)
            tr.barrier()
            with tr.mixed_region():
                state["umc"] = [u[j] - tr.xch_get(slots.c[j]) for j in range(n)]
                state["acc"] = [tr.zero() for _ in range(n)]

        def on_final(j, k, m):
            slot = slots.minv.get((j, k))
            if slot is None or not mine(k):
                return
            tr.xch_put(slot, m)
            with tr.mixed_region():
                state["acc"][j] = tr.fma(m, state["umc"][k], state["acc"][j])
                if k != j:
                    state["acc"][k] = tr.fma(m, state["umc"][j], state["acc"][k])
        alg.direct_minv(tr, spec, X, I, between=between, on_final=on_final)
        for j in range(n):
            tr.xch_put(out_slots[j], state["acc"][j])
        tr.barrier()
    else:
        if role == "consumer_c" or pre_barrier is None:
            c = alg.rnea(tr, spec, X, I, qd, None, g)[0]
            for j in range(n):
                if role == "consumer_c":
                    tr.xch_put(slots.c[j], c[j])
                else:
                    tr.anchor(c[j])
        if pre_barrier is not None:
            pre_barrier()                # (consumers: work that needs neither Minv nor qdd, done while the producers are busy)
        tr.barrier()
        tr.barrier()
    if two:
        return [tr.xch_get(slots.qdd[j]) + tr.xch_get(slots.qdd2[j]) for j in range(n)]
    return [tr.xch_get(slots.qdd[j]) for j in range(n)]


def exchange_dependence(tr):
    """dep[k]: node k depends (transitively) on a value read from the exchange region (i.e. on another wave's result)."""
    dep = [False] * len(tr.nodes)
    for k in range(1, len(tr.nodes)):
        op, a = tr.nodes[k][0], tr.nodes[k][1]
        dep[k] = str(a).startswith("in.xch_get(") if op == "in" else any(dep[d] for d in tr._deps(k))
    return dep


def coop_phase_costs(tr):
    """(instructions that can be issued before the first barrier, instructions that need another wave's result): arithmetic
    of the live part of a tile-cooperative core, split by exchange_dependence; exchange reads count with the second part."""
    dep = exchange_dependence(tr)
    live = tr.live_nodes()
    arith = ("fma", "mul", "add", "pkfma", "pkmul", "pkadd")
    p1 = sum(1 for k in range(1, len(tr.nodes)) if live[k] and not dep[k] and tr.nodes[k][0] in arith)
    p2 = sum(1 for k in range(1, len(tr.nodes)) if live[k] and dep[k] and (tr.nodes[k][0] in arith or tr.nodes[k][0] == "in"))
    return p1, p2


def hoist_before_first_barrier(tr):
    """Demand-order cores only.  Everything a core needs that does NOT depend on the exchange region is forced in front of its
    first barrier: the frontier of that independent part (values whose users all wait for another wave) is anchored there.
    While the producer wave runs the Minv recursion the other waves then do ~3/4 of their gradient work (dv, the whole d/dqd
    half, every velocity product) instead of waiting; what is left behind the barriers is only what depends on qdd."""
    dep = exchange_dependence(tr)
    live = tr.live_nodes()
    users = [[] for _ in tr.nodes]
    for k in range(1, len(tr.nodes)):
        if live[k]:
            for d in tr._deps(k):
                users[d].append(k)
    first = next(i for i, (dst, _) in enumerate(tr.outputs) if dst == "barrier")
    later_direct = set(abs(r) for (dst, r) in tr.outputs[first:] if not isinstance(r, float))
    frontier = [k for k in range(1, len(tr.nodes)) if live[k] and not dep[k] and tr.nodes[k][0] != "in"
                and (any(dep[u] for u in users[k]) or k in later_direct)]
    pos = tr.out_pos[first]
    tr.outputs[first:first] = [("anchor", k) for k in frontier]
    tr.out_pos[first:first] = [pos] * len(frontier)
    return len(frontier)


def core_forward_dynamics_gradient_coop(spec, role, cols, slots, hoist=False):
    """Fused forward-dynamics-gradient core of ONE wave of a tile-cooperative block (small robots).  The block's waves share
    the prefix instead of repeating it (column-split kernels): one wave computes Minv and qdd while the others compute the bias
    torques; results cross through LDS; then every wave differentiates its own columns.  cols may be empty (a producer without
    columns).  hoist=True (experimental, measured slower: 12.9 vs 12.1 us for iiwa-7 at K = 16384) additionally forces every
    part of a consumer's columns that does not depend on qdd in front of the first barrier (hoist_before_first_barrier).
    Same arithmetic per value as core_forward_dynamics_gradient."""
    assert role in COOP_ROLES
    n = spec.n
    tr, ins, g, X, I = _setup(spec, ["q", "qd", "u"])
    qdd = _coop_prologue(tr, spec, slots, role, X, I, ins["qd"], ins["u"], g)
    if not cols:
        return tr
    Mx = {}
    for (r, k), slot in slots.minv.items():
        Mx[(r, k)] = tr.xch_get(slot)
    Minv = [[(Mx.get((r, k), tr.zero()) if r <= k else None) for k in range(n)] for r in range(n)]
    c2, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], qdd, g)
    dc = alg.rnea_grad(tr, spec, X, I, ins["qd"], v, a, f, g)
    out = alg.fd_grad_finish(tr, spec, Minv, dc, cols)
    _out_grad(tr, spec, out, cols)
    if hoist and role != "producer":
        hoist_before_first_barrier(tr)
    return tr


def rollout_row_count(spec):
    """Values per configuration and step of the rollout output: [x+ (2n) | A (2n x 2n, column-major) | B (2n x n)]."""
    return 2 * spec.n * (1 + 3 * spec.n)


class RolloutChunks:
    """Bookkeeping of the rollout step cores: the output row is written in chunks of 2n values (x+, then one column of A or
    B each); a core may emit the chunks in any order -- chunk k of the emission goes to row offset bases[k]."""

    def __init__(self, tr, n):
        self.tr, self.n, self.bases = tr, n, []

    def put(self, base, values):
        assert len(values) == 2 * self.n
        k = len(self.bases)
        self.bases.append(base)
        for i, v in enumerate(values):
            self.tr.out(2 * self.n * k + i, v)

    def x_next(self, q, qd, qdd, dt):
        n = self.n
        qdn = [qd[j] + dt * qdd[j] for j in range(n)]               # semi-implicit Euler: velocity first,
        qn = [q[j] + dt * qdn[j] for j in range(n)]                 # then the position with the NEW velocity
        self.put(0, qn + qdn)

    def B_columns(self, Minv, dt, dt2):
        n = self.n
        for c in range(n):
            col = [alg.minv_sym(Minv, r, c) for r in range(n)]
            self.put(2 * n + 4 * n * n + 2 * n * c, [dt2 * e for e in col] + [dt * e for e in col])

    def A_columns(self, col, lo, hi, dt, dt2):
        """lo = dqdd/dq[:, col], hi = dqdd/dqd[:, col] -> columns col and n + col of A."""
        n = self.n
        one = lambda r: 1.0 if r == col else 0.0
        self.put(2 * n + 2 * n * col, [dt2 * lo[r] + one(r) for r in range(n)] + [dt * lo[r] for r in range(n)])
        self.put(2 * n + 2 * n * (n + col), [dt2 * hi[r] + dt * one(r) for r in range(n)] + [dt * hi[r] + one(r) for r in range(n)])


def core_rollout_step(spec):
    """One semi-implicit Euler step of the forward dynamics with its linearisation (fused trace, small robots):
    in: q, qd, u, dt -> out row [x+ | A | B] (oracle/rbd_oracle.py: rollout_step).  Returns (tracer, chunk bases)."""
    n = spec.n
    tr, ins, g, X, I = _setup(spec, ["q", "qd", "u"])
    dt = tr.inp("in.dt()")
    dt2 = dt * dt
    Minv = alg.direct_minv(tr, spec, X, I)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], None, g)
    qdd = alg.fd_finish(tr, spec, Minv, ins["u"], c)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], qdd, g)
    dc = alg.rnea_grad(tr, spec, X, I, ins["qd"], v, a, f, g)
    df = alg.fd_grad_finish(tr, spec, Minv, dc)
    Minv32 = alg.minv_in_compute_type(tr, Minv)
    ch = RolloutChunks(tr, n)
    ch.x_next(ins["q"], ins["qd"], qdd, dt)
    for col in range(n):
        ch.A_columns(col, [df[r][col].lo for r in range(n)], [df[r][col].hi for r in range(n)], dt, dt2)
    ch.B_columns(Minv32, dt, dt2)
    order = sorted(range(len(ch.bases)), key=lambda k: ch.bases[k])
    assert [ch.bases[k] for k in order] == [2 * n * k for k in range(1 + 3 * n)]
    return tr, list(ch.bases)


def _arith_ops(tr):
    return tr.arith_instructions()


def range_cost_function(spec, builder, exact):
    """cost(b, e): arithmetic ops of a part holding columns b..e-1.  exact: trace every range that is asked for
    (small robots); otherwise a shared prefix + per-column marginal model from n single-column traces."""
    memo = {}
    if exact:
        def cost(b, e):
            if (b, e) not in memo:
                memo[(b, e)] = _arith_ops(builder(list(range(b, e))))
            return memo[(b, e)]
        return cost
    single = [_arith_ops(builder([c])) for c in range(spec.n)]
    prefix = min(single)
    pre = [0]
    for x in single:
        pre.append(pre[-1] + max(0, x - prefix))
    return lambda b, e: prefix + pre[e] - pre[b]


def balanced_column_split(spec, S, cost, first=0):
    """Split columns first..n-1 into S contiguous ranges minimising the largest part's operation count.
    Returns (parts, ops of the largest part according to `cost`)."""
    n = spec.n
    S = max(1, min(S, n - first))
    INF = float("inf")
    best = [[INF] * (n + 1) for _ in range(S + 1)]
    cut = [[0] * (n + 1) for _ in range(S + 1)]
    best[0][first] = 0
    for k in range(1, S + 1):
        for e in range(first + k, n + 1):
            for b in range(first + k - 1, e):
                if best[k - 1][b] == INF:
                    continue
                c = max(best[k - 1][b], cost(b, e))
                if c < best[k][e]:
                    best[k][e] = c
                    cut[k][e] = b
    bounds = [n]
    e = n
    for k in range(S, 0, -1):
        e = cut[k][e]
        bounds.append(e)
    bounds = bounds[::-1]
    return [list(range(bounds[i], bounds[i + 1])) for i in range(S)], best[S][n]

def refine_contiguous_split(parts, builder, max_steps=8, per_column=0):
    """Hill-climb a contiguous column split on EXACT group costs (each evaluation traces a group): move one boundary column
    out of the most expensive group into its neighbour while that lowers the maximum.  Used where the prefix + marginal
    model of range_cost_function is poor (large robots, dFD: the Minv rows a group needs depend on its columns; Atlas-30
    x4: 25.5 k -> about 21 k operations in the largest group).  Returns (parts, exact ops of the largest group)."""
    parts = [list(p) for p in parts]
    memo = {}

    def cost(p):
        key = (p[0], p[-1])
        if key not in memo:
            memo[key] = _arith_ops(builder(list(p))) + per_column * len(p)      # (+ what flushing the columns costs a lone wave)
        return memo[key]

    for _ in range(max_steps):
        costs = [cost(p) for p in parts]
        worst = max(range(len(parts)), key=lambda i: costs[i])
        best = None
        for nb, col_from_end in ((worst - 1, False), (worst + 1, True)):
            if nb < 0 or nb >= len(parts) or len(parts[worst]) < 2:
                continue
            if col_from_end:       # give the last column to the right neighbour
                a, b = parts[worst][:-1], [parts[worst][-1]] + parts[nb]
            else:                  # give the first column to the left neighbour
                a, b = parts[worst][1:], parts[nb] + [parts[worst][0]]
            new_max = max([cost(a), cost(b)] + [costs[i] for i in range(len(parts)) if i not in (worst, nb)])
            if new_max < costs[worst] and (best is None or new_max < best[0]):
                best = (new_max, nb, a, b)
        if best is None:
            break
        _, nb, a, b = best
        parts[worst], parts[nb] = a, b
    return parts, max(cost(p) for p in parts)


def optimal_column_sets(spec, S, full):
    """Best partition of the n gradient columns into S arbitrary SETS (n <= 8: exhaustive over all set partitions).
    `full` is the trace of the whole gradient (outputs at n*col + row and n*n + n*col + row); the cost of a set is the number
    of arithmetic nodes alive when only its columns are kept -- exactly what tracing that group would give, without
    re-tracing.  Early columns are several times as expensive as late ones (d/dq_1 touches every descendant), so for a
    7-joint chain {0,4}, {1}, {2,6}, {3,5} (2937 ops, = column 1 alone) beats the best contiguous split
    {0,1}, {2}, {3}, {4,5,6} (3580).  Returns (parts sorted by first column, ops of the largest part)."""
    n = spec.n
    assert n <= 8 and 1 <= S <= n
    roots = {c: [] for c in range(n)}
    for (dst, ref) in full.outputs:
        if not isinstance(dst, str) and not isinstance(ref, float):
            roots[(int(dst) % (n * n)) // n].append(abs(ref))
    arith = ("fma", "mul", "add", "pkfma", "pkmul", "pkadd")
    memo = {}

    def cost(mask):
        if mask not in memo:
            live = [False] * len(full.nodes)
            stack = [r for c in range(n) if mask >> c & 1 for r in roots[c]]
            while stack:
                k = stack.pop()
                if live[k]:
                    continue
                live[k] = True
                stack.extend(d for d in full._deps(k) if not live[d])
            memo[mask] = sum(1 for k in range(1, len(full.nodes)) if live[k] and full.nodes[k][0] in arith)
        return memo[mask]

    best = [None, None]

    def rec(i, masks):
        if i == n:
            if len(masks) == S:
                worst = max(cost(m) for m in masks)
                if best[0] is None or worst < best[0]:
                    best[0], best[1] = worst, list(masks)
            return
        if len(masks) + (n - i) < S:
            return
        for b in range(len(masks)):
            masks[b] |= 1 << i
            if best[0] is None or cost(masks[b]) < best[0]:       # the largest group can only grow
                rec(i + 1, masks)
            masks[b] &= ~(1 << i)
        if len(masks) < S:
            masks.append(1 << i)
            rec(i + 1, masks)
            masks.pop()

    rec(0, [])
    parts = sorted([[c for c in range(n) if m >> c & 1] for m in best[1]], key=lambda p: p[0])
    return parts, best[0]


WAVE_LANES = 64
FLUSH_FIXED_SLOTS = 40.0        # wave syncs, address set-up and the LDS round trips of one flush, in issue slots


def optimal_half_column_sets(spec, S, full, restarts=20, weights=None, per_value=0.0):
    """Partition of the 2n HALF columns -- (column, d/dq) and (column, d/dqd) are separate items: the two halves of a gradient column
    share nothing but the prefix -- into S groups minimising the largest group's arithmetic (cost of a group = live arithmetic nodes
    of `full` when only its outputs are kept, as in optimal_column_sets).  2n items are too many for the exhaustive search; this is
    longest-processing-time placement followed by single moves and swaps out of the heaviest group, from `restarts` deterministic
    perturbed orders.  iiwa-7 forward-dynamics gradient, S = 4: 2771 operations against 2937 for whole columns.
    per_value: issue slots charged per OUTPUT value of a group (staging write, flush read, store: the stamped 4-way kernel's groups took
    15.3 / 15.3 / 13.9 / 13.9 k cycles for 2804 / 2805 / 2700 / 2718 operations and 35 / 35 / 14 / 14 values -- ~43 cycles = ~10 slots per
    value, profiles/r03/phase_stamps_iiwa7_split4_16384.txt).  weights[g]: the cost of group g counts weights[g]-fold -- groups that run
    as the YOUNGER wave of a SIMD get what the older one leaves (asymmetric 8-way split: four heavy groups dispatched first, four
    light ones behind them); groups keep their index then (no sorting).
    Returns (parts, worst) with parts = [(d/dq columns, d/dqd columns)] sorted by their first column (unweighted) / in group order."""
    import random
    n = spec.n
    roots = {}
    for (dst, ref) in full.outputs:
        if not isinstance(dst, str) and not isinstance(ref, float):
            i = int(dst)
            roots.setdefault(((i % (n * n)) // n, i // (n * n)), []).append(abs(ref))
    arith = ("fma", "mul", "add", "pkfma", "pkmul", "pkadd")
    is_arith = [False] + [full.nodes[k][0] in arith for k in range(1, len(full.nodes))]
    bits = {}
    for item in [(c, h) for c in range(n) for h in (0, 1)]:
        live, stack = set(), list(roots.get(item, []))
        while stack:
            k = stack.pop()
            if k not in live:
                live.add(k)
                stack.extend(d for d in full._deps(k) if d not in live)
        mask = 0
        for k in live:
            if is_arith[k]:
                mask |= 1 << k
        bits[item] = mask
    memo = {}

    def cost0(group):
        key = frozenset(group)
        if key not in memo:
            m = 0
            for it in key:
                m |= bits[it]
            if per_value == "flush":
                # what the two flushes of a group (grid_out_colset2: its d/dq values, then its d/dqd values) execute: LEN staging writes,
                # then GRID_WAVE_SIZE / G iterations of (LDS read + store) with G = 64 / pow2ceil(LEN) configurations per iteration --
                # 14 values: 16 iterations, 21 or 28: 32, 35: 64 (29 of 64 lanes idle) -- plus the wave syncs and address set-up
                extra = 0.0
                for half in (0, 1):
                    length = n * sum(1 for (_, h) in key if h == half)
                    if length:
                        p2 = 1
                        while p2 < length:
                            p2 *= 2
                        extra += length + 2.0 * (WAVE_LANES // max(1, WAVE_LANES // p2)) + FLUSH_FIXED_SLOTS
                memo[key] = bin(m).count("1") + extra
            else:
                memo[key] = bin(m).count("1") + per_value * n * len(key)
        return memo[key]
    wts = [1.0] * S if weights is None else list(weights)
    gcost = lambda k, group: (cost0(group) * wts[k]) if group else 0.0
    cost = cost0
    items = sorted(bits)
    best = None
    for trial in range(restarts):
        rng = random.Random(trial)
        order = sorted(items, key=lambda it: -cost([it]) + rng.random() * (0 if trial == 0 else 300))
        groups = [[] for _ in range(S)]
        for it in order:
            k = min(range(S), key=lambda k: (gcost(k, groups[k] + [it]), k))
            groups[k].append(it)
        improved = True
        while improved:
            improved = False
            cs = [gcost(k, gp) for k, gp in enumerate(groups)]
            w = max(range(S), key=lambda i: cs[i])
            moves = []
            for it in groups[w]:
                for k in range(S):
                    if k == w:
                        continue
                    moves.append((it, k, None))
                    moves.extend((it, k, jt) for jt in groups[k])
            for (it, k, jt) in moves:
                ng = [list(gp) for gp in groups]
                ng[w].remove(it); ng[k].append(it)
                if jt is not None:
                    ng[k].remove(jt); ng[w].append(jt)
                if all(ng) and max(gcost(k, gp) for k, gp in enumerate(ng)) < cs[w]:
                    groups, improved = ng, True
                    break
        worst = max(gcost(k, gp) for k, gp in enumerate(groups))
        if all(groups) and (best is None or worst < best[0]):
            best = (worst, [sorted(gp) for gp in groups])
    parts = [(sorted(c for (c, h) in gp if h == 0), sorted(c for (c, h) in gp if h == 1)) for gp in best[1]]
    if weights is None:
        parts.sort(key=lambda p: min(p[0] + p[1]))
    return parts, best[0]


def balanced_column_sets(spec, S, builder):
    """Like balanced_column_split but the groups may be arbitrary column SETS: longest-processing-time assignment on the
    single-column costs (shared prefix + marginal), then the exact cost of each group by tracing it.  Early columns are
    several times as expensive as late ones (d/dq_1 touches every descendant), so {1}, {2,0}, {3,5}, {4,6} beats the best
    contiguous split {0,1}, {2}, {3}, {4,5,6} of a 7-joint chain.  Returns (parts, ops of the largest part)."""
    n = spec.n
    single = [_arith_ops(builder([c])) for c in range(n)]
    prefix = min(single)
    bins = [[] for _ in range(S)]
    load = [0.0] * S
    for c in sorted(range(n), key=lambda c: -single[c]):
        k = min(range(S), key=lambda k: (load[k], k))
        bins[k].append(c)
        load[k] += single[c] - prefix
    parts = [sorted(b) for b in bins if b]
    parts.sort(key=lambda p: p[0])
    return parts, max(_arith_ops(builder(p)) for p in parts)


# ------------------------------------------------------------------------------------------------
# pointer-style ``_inner`` bodies (API parity with the reference's ALGORITHM_inner tier)
# ------------------------------------------------------------------------------------------------
def inner_inverse_dynamics(spec, compute_c, use_qdd):
    """inverse_dynamics_inner / inverse_dynamics_inner_vaf: writes s_vaf (and s_c)."""
    tr, ins, g, X, I = _setup(spec, ["q", "qd"] + (["qdd"] if use_qdd else []), "inner")
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], ins.get("qdd"), g)
    n = spec.n
    if compute_c:
        for j in range(n):
            tr.out("s_c[%d]" % j, c[j])
    for blk, arr in enumerate((v, a, f)):
        for j in range(n):
            for r in range(6):
                tr.out("s_vaf[%d]" % (6 * n * blk + 6 * j + r), arr[j][r])
    return tr


def inner_direct_minv(spec):
    tr, ins, g, X, I = _setup(spec, ["q"], "inner")
    Minv = alg.direct_minv(tr, spec, X, I)
    n = spec.n
    for col in range(n):
        for row in range(n):
            tr.out("s_Minv[%d]" % (n * col + row), Minv[row][col] if row <= col else tr.zero())
    return tr


def inner_forward_dynamics_finish(spec):
    tr = Tracer()
    n = spec.n
    u = [tr.inp("s_u[%d]" % j) for j in range(n)]
    c = [tr.inp("s_c[%d]" % j) for j in range(n)]
    Min = [tr.inp("s_Minv[%d]" % i) for i in range(n * n)]
    Minv = [[Min[n * cc + r] if r <= cc else None for cc in range(n)] for r in range(n)]
    qdd = alg.fd_finish(tr, spec, Minv, u, c)
    for j in range(n):
        tr.out("s_qdd[%d]" % j, qdd[j])
    return tr


def inner_forward_dynamics(spec):
    tr, ins, g, X, I = _setup(spec, ["q", "qd", "u"], "inner")
    Minv = alg.direct_minv(tr, spec, X, I)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], None, g)
    qdd = alg.fd_finish(tr, spec, Minv, ins["u"], c)
    for j in range(spec.n):
        tr.out("s_qdd[%d]" % j, qdd[j])
    return tr


def inner_inverse_dynamics_gradient_columns(spec):
    """inverse_dynamics_gradient_inner for large robots: the column-serial walk of rnea_grad_columns over s_vaf (v, a, f are
    re-read per column, X_j(q) is rematerialised per column from the lane's sin/cos table), creation-order emission.  The
    fused trace below keeps ~900 values alive for a 30-joint robot and its GPU build spilled kilobytes per lane."""
    n = spec.n
    tr = Tracer()
    q = [tr.inp("s_q[%d]" % j) for j in range(n)]
    qd = [tr.inp("s_qd[%d]" % j) for j in range(n)]
    g = tr.inp("gravity")
    trig = [(tr.inp("s_XImats[%d]" % j), tr.inp("s_XImats[%d]" % (n + j))) if spec.uses_trig[j] else None for j in range(n)]
    I = alg.build_I(tr, spec)
    serial = [0]

    def load6(base, j):
        serial[0] += 1
        return [tr.inp("s_vaf[%d]/*%d*/" % (base + 6 * j + r, serial[0])) for r in range(6)]

    def loader(kind, j):
        if kind == "v":
            return load6(0, j)
        if kind == "f":
            return load6(12 * n, j)
        Xj = alg.build_X_joint(tr, spec, j, q[j], trig[j])          # "xa": X_j a_parent (base: X_j[:, 5] g)
        p = spec.parent[j]
        return alg.matvec(tr, Xj, load6(6 * n, p)) if p != -1 else [Xj[r][5] * g for r in range(6)]

    def emit_column(col, dc):
        for half in (0, 1):
            for r in range(n):
                e = dc.get(r)
                tr.out("s_dc_du[%d]" % (half * n * n + n * col + r), e[half] if e is not None else tr.zero())

    alg.rnea_grad_columns(tr, spec, I, q, qd, trig, loader, emit_column, prefetch=0)
    return tr


def inner_inverse_dynamics_gradient(spec):
    """inverse_dynamics_gradient_inner: v, a, f come in through s_vaf (as in the reference)."""
    tr, ins, g, X, I = _setup(spec, ["q", "qd"], "inner")
    n = spec.n
    v = [[tr.inp("s_vaf[%d]" % (6 * j + r)) for r in range(6)] for j in range(n)]
    a = [[tr.inp("s_vaf[%d]" % (6 * n + 6 * j + r)) for r in range(6)] for j in range(n)]
    f = [[tr.inp("s_vaf[%d]" % (12 * n + 6 * j + r)) for r in range(6)] for j in range(n)]
    dc = alg.rnea_grad(tr, spec, X, I, ins["qd"], v, a, f, g)
    for col in range(n):
        for row in range(n):
            tr.out("s_dc_du[%d]" % (n * col + row), dc[row][col].lo)
    for col in range(n):
        for row in range(n):
            tr.out("s_dc_du[%d]" % (n * n + n * col + row), dc[row][col].hi)
    return tr


# ------------------------------------------------------------------------------------------------
# two-pass "pipeline" cores for robots whose gradient working set exceeds the register file
# ------------------------------------------------------------------------------------------------
class WorkspaceMap:
    """Slot numbering of the per-configuration workspace shared by a prep core and a columns core.
    Layout in memory is tile-major SoA: value `slot` of lane `l` of tile `t` lives at (t*count + slot)*64 + l, so every
    access of a wavefront is one coalesced 256-byte transaction."""

    def __init__(self, spec, with_minv):
        n = spec.n
        self.n = n
        self.V = 0                  # v_j            6n
        self.XA = 6 * n             # X_j a_parent   6n
        self.F = 12 * n             # accumulated f  6n
        self.SC = 18 * n            # sin q, cos q   2n
        self.count = 20 * n
        self.minv_slot = {}
        if with_minv:
            nz = alg.minv_zero_pattern(spec)
            for r in range(n):
                for k in range(r, n):
                    if nz[r][k]:
                        self.minv_slot[(r, k)] = self.count
                        self.count += 1

    def minv(self, r, k):
        return self.minv_slot.get((r, k) if r <= k else (k, r))


def _xa_from_rnea(tr, spec, X, a, g):
    out = []
    for j in range(spec.n):
        p = spec.parent[j]
        out.append(alg.matvec(tr, X[j], a[p]) if p != -1 else [X[j][r][5] * g for r in range(6)])
    return out


def _out_workspace(tr, spec, ws, trig, v, xa, f, Minv=None):
    n = spec.n
    for j in range(n):
        for r in range(6):
            tr.out(ws.V + 6 * j + r, v[j][r])
            tr.out(ws.XA + 6 * j + r, xa[j][r])
            tr.out(ws.F + 6 * j + r, f[j][r])
        s, c = trig[j] if trig[j] is not None else (tr.zero(), tr.const(1.0))
        tr.out(ws.SC + j, s)
        tr.out(ws.SC + n + j, c)
    if Minv is not None:
        for (r, k), slot in ws.minv_slot.items():
            tr.out(slot, Minv[r][k])


def core_gradient_prep(spec, ws, kind, use_qdd=False):
    """Pass 1.  kind "id": RNEA at (q, qd[, qdd]); kind "fd": Minv, RNEA(0), qdd = Minv (u - c), RNEA(qdd).
    Writes v, X a_parent, accumulated f, sin/cos (and the non-zero upper triangle of Minv) to the workspace."""
    if kind == "id":
        tr, ins, g, X, I = _setup(spec, ["q", "qd"] + (["qdd"] if use_qdd else []))
        trig = alg.trig_from_q(tr, spec, ins["q"])
        c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], ins.get("qdd"), g)
        _out_workspace(tr, spec, ws, trig, v, _xa_from_rnea(tr, spec, X, a, g), f)
        return tr
    tr, ins, g, X, I = _setup(spec, ["q", "qd", "u"])
    trig = alg.trig_from_q(tr, spec, ins["q"])
    Minv = alg.direct_minv(tr, spec, X, I)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], None, g)
    qdd = alg.fd_finish(tr, spec, Minv, ins["u"], c)
    c, v, a, f = alg.rnea(tr, spec, X, I, ins["qd"], qdd, g)
    _out_workspace(tr, spec, ws, trig, v, _xa_from_rnea(tr, spec, X, a, g), f, Minv)
    return tr


def core_gradient_columns(spec, ws, with_minv, prefetch=3):
    """Pass 2 (creation-order emission).  Column-serial gradient reading the workspace; with_minv also applies
    df_du[:, col] = -Minv_sym dc_du[:, col] per column.  Outputs: column `col` of d/dq (indices n*col..) then of d/dqd."""
    n = spec.n
    tr = Tracer()
    q = [tr.inp("in.q(%d)" % j) for j in range(n)]
    qd = [tr.inp("in.qd(%d)" % j) for j in range(n)]
    trig = [(tr.inp("in.ws(%d)" % (ws.SC + j)), tr.inp("in.ws(%d)" % (ws.SC + n + j))) if spec.uses_trig[j] else None
            for j in range(n)]
    I = alg.build_I(tr, spec)
    base = {"v": ws.V, "xa": ws.XA, "f": ws.F}
    serial = [0]

    def loader(kind, j):
        # a fresh load node per request ("#k" keeps requests of the same slot distinct: each is its own global_load)
        serial[0] += 1
        return [tr.inp("in.ws(%d)/*%d*/" % (base[kind] + 6 * j + r, serial[0])) for r in range(6)]

    def emit_column(col, dc):
        rows = sorted(dc)
        if not with_minv:
            for half in (0, 1):
                for r in range(n):
                    e = dc.get(r)
                    tr.out(half * n * n + n * col + r, e[half] if e is not None else tr.zero())
            return
        # -Minv_sym dc: both halves of a row share the Minv loads; the second half waits in registers for its chunk
        hi = []
        tr.fence()
        for r in range(n):
            m = [(k, tr.inp("in.ws(%d)" % ws.minv(r, k))) for k in rows if ws.minv(r, k) is not None]
            tr.out(n * col + r, -tr.dot([(mk, dc[k][0]) for (k, mk) in m]))
            hi.append(-tr.dot([(mk, dc[k][1]) for (k, mk) in m]))
            if r % 3 == 2:
                tr.fence()
        for r in range(n):
            tr.out(n * n + n * col + r, hi[r])

    alg.rnea_grad_columns(tr, spec, I, q, qd, trig, loader, emit_column, prefetch=prefetch)
    return tr


# ------------------------------------------------------------------------------------------------
# single-kernel column-serial cores with RECOMPUTATION (large robots)
# ------------------------------------------------------------------------------------------------
def recompute_table_size(spec, kind, use_qdd=False, use_qdd_minv=False):
    """Entries of the lane-private table of core_gradient_recompute(table=True): sin, cos, qd [, qdd] [, q if prismatic]."""
    has_qdd = (kind == "fd") or use_qdd
    return spec.n * (3 + (1 if has_qdd else 0) + (1 if any(not t for t in spec.uses_trig) else 0))


def core_gradient_recompute(spec, kind, use_qdd=False, use_qdd_minv=False, table=False, facc_separate=True, cols=None, rollout=None,
                            coop=None):
    """Column-serial gradient core that keeps almost nothing alive between columns: inside each column the velocities,
    accelerations and accumulated forces it needs (path root -> column joint, and the column's subtree) are RECOMPUTED
    from q, qd, qdd instead of being held in registers (the fused demand-ordered trace keeps 880-1160 values alive for
    Atlas-30 and spills) or parked in an HBM workspace (two-pass variant).  ~1.6x the arithmetic, no spills.

    kind "id": dc_du at (q, qd[, qdd]).  kind "fd": df_du; Minv and qdd are computed first (default) or read from the
    inputs (use_qdd_minv: the reference's USE_QDD_MINV_FLAG variant; Minv entries are read where they are used).

    coop = (role, CoopSlots): the core of one wave of a tile-cooperative block (kind "fd"): Minv is produced by one wave and
    read from the block's LDS exchange region where it is used (never held in registers: the 465 values of Atlas-30 are what
    made every dFD column group spill), see core_forward_dynamics_gradient_coop.

    rollout: a list that receives the chunk bases -- the core then is one semi-implicit Euler step with its linearisation
    (kind "fd" only; see core_rollout_step): x+ and the B columns are emitted after the prologue, two A columns per gradient
    column.

    table=True: sin q, cos q, qd, qdd are parked in a lane-private table after the prologue (tab_put) and re-loaded per
    column (tab_get) instead of staying in ~4n registers for the whole kernel: with them resident the Atlas-30 kernels
    spill 200-600 values to scratch, and once more than ~512 waves run the scratch lines no longer stay in L2 (measured:
    dID 167 us at K=32768, 300 us at K=49152)."""
    n = spec.n
    tr = Tracer()
    mark0 = tr.cse_mark()
    q = [tr.inp("in.q(%d)" % j) for j in range(n)]
    qd = [tr.inp("in.qd(%d)" % j) for j in range(n)]
    g = tr.inp("gravity")
    trig = alg.trig_from_q(tr, spec, q)
    I = alg.build_I(tr, spec)
    Minv = None
    lean = coop is not None and coop[1].lean
    itab = coop[1].itab if lean else None
    want = None             # lean cores: cols = [(column, half)] -> {column: halves to emit}; tr.run_bases = row offset of each emitted run
    if coop is not None and cols and isinstance(cols[0], tuple):
        want = {}
        for (c_, h_) in cols:
            want.setdefault(c_, set()).add(h_)
        cols = sorted(want)
    tr.run_bases = []
    pass_halves = []        # (lean cores with separate_halves: the halves each pass of the column loop emits, in order)
    stream = None           # (lean cores with aligned_flush: AlignedPieces, set once the order of the half-columns is known)
    pairing = {"role": [], "at": 0, "held": None}      # (lean cores with pair_products: per pass "first" | "second" | None; the held first column)
    if coop is not None:
        assert (kind == "fd" or lean) and not use_qdd_minv and not table and rollout is None
        role, slots = coop
        qdd = None          # (the prologue runs further down, once the per-column helpers it may call are defined)
    elif kind == "fd" and not use_qdd_minv:
        mark = tr.cse_mark()
        u = [tr.inp("in.u(%d)" % j) for j in range(n)]
        X = alg.build_X(tr, spec, q, trig)
        Minv = alg.direct_minv(tr, spec, X, I)
        c = alg.rnea(tr, spec, X, I, qd, None, g)[0]
        qdd = alg.fd_finish(tr, spec, Minv, u, c)
        qdd = list(qdd)
        Minv = alg.minv_in_compute_type(tr, Minv)      # mixed precision: only the rounded copies stay alive below
        if rollout is not None:
            dt = tr.inp("in.dt()")
            dt2 = dt * dt
            chunks = RolloutChunks(tr, n)
            chunks.bases = rollout                     # (the caller's list)
            chunks.x_next(q, qd, qdd, dt)
            chunks.B_columns(Minv, dt, dt2)
        tr.fence()
        tr.cse_release(mark, keep=[t.ref for pair in trig if pair is not None for t in pair])
    elif kind == "fd" or use_qdd:
        qdd = [tr.inp("in.qdd(%d)" % j) for j in range(n)]
    else:
        qdd = None
    nz = alg.minv_zero_pattern(spec) if kind == "fd" else None

    memo = {}
    saved_dqd = {}
    trig = list(trig)
    if table:
        slot = {"s": 0, "c": n, "qd": 2 * n}
        nxt = 3 * n
        if qdd is not None:
            slot["qdd"] = nxt
            nxt += n
        if any(not t for t in spec.uses_trig):
            slot["q"] = nxt
            nxt += n
        assert nxt == recompute_table_size(spec, kind, use_qdd, use_qdd_minv)
        for j in range(n):
            if trig[j] is not None:
                tr.tab_put(slot["s"] + j, trig[j][0])
                tr.tab_put(slot["c"] + j, trig[j][1])
            elif "q" in slot:
                tr.tab_put(slot["q"] + j, q[j])
            tr.tab_put(slot["qd"] + j, qd[j])
            if qdd is not None:
                tr.tab_put(slot["qdd"] + j, qdd[j])
        tr.fence()
        tr.cse_release(mark0)       # nothing of the prologue is reused by name below (Minv entries are held by reference)

    def touch(j):
        # first use of joint j in this column: launder its inputs IN PLACE, so the chains recomputed below are new values
        # to the compiler (otherwise its CSE keeps the first column's v, a, f alive for all later columns and spills them)
        if ("t", j) not in memo and table:
            memo[("t", j)] = True
            if trig[j] is not None:
                trig[j] = (tr.tab_get(slot["s"] + j), tr.tab_get(slot["c"] + j))
            else:
                q[j] = tr.tab_get(slot["q"] + j)
            qd[j] = tr.tab_get(slot["qd"] + j)
            if qdd is not None:
                qdd[j] = tr.tab_get(slot["qdd"] + j)
        elif ("t", j) not in memo and itab is not None:
            # block-shared input table (CoopSlots.enable_lean): a fresh LDS read per touch, nothing of the inputs stays in registers
            memo[("t", j)] = True
            if trig[j] is not None:
                trig[j] = (tr.xch_get(itab["s"][j]), tr.xch_get(itab["c"][j]))
            else:
                q[j] = tr.xch_get(itab["q"][j])
            qd[j] = tr.xch_get(itab["qd"][j])
            if qdd is not None:
                qdd[j] = tr.xch_get(coop[1].qdd[j])
        elif ("t", j) not in memo:
            memo[("t", j)] = True
            q[j] = tr.launder(q[j])
            qd[j] = tr.launder(qd[j])
            if trig[j] is not None:
                trig[j] = tuple(tr.launder(t) for t in trig[j])
            if qdd is not None:
                qdd[j] = tr.launder(qdd[j])

    def Xof(j):
        if ("X", j) not in memo:
            touch(j)
            memo[("X", j)] = alg.build_X_joint(tr, spec, j, q[j], trig[j])
        return memo[("X", j)]

    def Xof_back(j):
        # lean cores: X_j for the force transfer child -> parent is REBUILT (from re-read sin / cos) instead of staying alive across
        # the child's whole subtree -- unless that subtree is so small (CoopSlots.keep_x_below joints) that only the last levels of a
        # path hold their X at once
        if len(spec.subtree[j]) <= getattr(coop[1], "keep_x_below", 0) and ("X", j) in memo:
            return memo[("X", j)]
        memo.pop(("X", j), None)
        memo.pop(("t", j), None)
        return Xof(j)

    def v_of(j):
        if ("v", j) not in memo:
            touch(j)
            p, s = spec.parent[j], spec.S_ind[j]
            if p == -1:
                v = alg.zeros6(tr)
                v[s] = qd[j]
            else:
                v = alg.matvec(tr, Xof(j), v_of(p))
                v[s] = v[s] + qd[j]
            memo[("v", j)] = v
        return memo[("v", j)]

    def xa_of(j):
        if ("xa", j) not in memo:
            p = spec.parent[j]
            memo[("xa", j)] = alg.matvec(tr, Xof(j), a_of(p)) if p != -1 else [Xof(j)[r][5] * g for r in range(6)]
        return memo[("xa", j)]

    def a_of(j):
        if ("a", j) not in memo:
            p, s = spec.parent[j], spec.S_ind[j]
            touch(j)
            a = list(xa_of(j))
            if p != -1:
                a = alg.vadd(a, alg.mxS(tr, s, v_of(j), qd[j]))
            if qdd is not None:
                a[s] = a[s] + qdd[j]
            memo[("a", j)] = a
        return memo[("a", j)]

    chain = {"f": None, "for": None}       # lean cores: (joint, its accumulated force) of the d/dq column finished last | the joint being accumulated
    ftab = {"on": False, "done": set(), "hook": None}     # tile-cooperative cores: f of the wave's own finished columns lives in LDS (CoopSlots.f)

    def facc(j):
        if ("f", j) not in memo and ftab["on"] and j in ftab["done"]:
            memo[("f", j)] = [tr.tab_get(coop[1].f[j] + r) for r in range(6)]
        if ("f", j) not in memo:
            v = v_of(j)
            f = alg.vadd(alg.matvec(tr, I[j], a_of(j)), alg.fxv(tr, v, alg.matvec(tr, I[j], v)))
            for ch in spec.children[j]:
                if lean:
                    # the accumulated force of the column this wave finished just before, when that is a child of this one (runs of
                    # d/dq columns are walked towards the root): six registers instead of the child's whole subtree once more
                    fch = chain["f"][1] if (chain["f"] is not None and chain["f"][0] == ch and j == chain["for"]) else facc(ch)
                    f = alg.mattvec_acc(tr, Xof_back(ch), fch, f)
                else:
                    f = alg.mattvec_acc(tr, Xof(ch), facc(ch), f)
            memo[("f", j)] = f
            if ftab["hook"] is not None:
                ftab["hook"](j, f)
        return memo[("f", j)]

    def loader(kind_, j):
        if kind_ == "f" and table and facc_separate and spec.children[j]:
            # the accumulated force of the column joint walks its whole subtree: do that on its own set of re-loaded
            # inputs and forget the v, a it produced (they are recomputed at the visits that use them), otherwise up to
            # 12 values per subtree joint stay alive from here to their visit
            saved = dict(memo)
            memo.clear()
            f = facc(j)
            memo.clear()
            memo.update(saved)
            for key in [key for key in memo if key[0] == "t"]:       # reload inputs for what follows
                del memo[key]
            for key in [key for key in memo if key[0] in ("X", "v", "xa", "a")]:
                del memo[key]
            return f
        if kind_ == "f" and lean:
            # the walk that accumulates the column joint's force visits the joint's whole subtree; what it computed on the way (v, a, X
            # of every joint below) is forgotten -- the gradient walk recomputes v where it needs it -- otherwise 6-12 values per subtree
            # joint stay alive from here to their visit (18 joints below the torso: 100+ registers)
            chain["for"] = j if getattr(coop[1], "chain_f", False) else None
            f = facc(j)
            chain["for"] = None
            below = set(spec.subtree[j]) - {j}
            for key in [key for key in memo if key[1] in below]:
                del memo[key]
            chain["f"] = (j, f)
            return f
        if kind_ == "f" and ftab["on"]:
            # The accumulated force of a column's joint walks the joint's whole subtree -- unless a child is a column this wave has
            # already finished (columns run deepest first): its f is then six LDS reads.  Atlas-30: a tenth of a tile's arithmetic.
            f = facc(j)
            if j not in ftab["done"] and j in coop[1].f:
                for r in range(6):
                    tr.tab_put(coop[1].f[j] + r, f[r])
                ftab["done"].add(j)
            return f
        return {"v": v_of, "xa": xa_of, "f": facc}[kind_](j)

    def minv_entry(r, k):
        a, b = (r, k) if r <= k else (k, r)
        if not nz[a][b]:
            return None
        if coop is not None:
            return coop[1].entry(tr, a, b)
        return Minv[a][b] if Minv is not None else tr.inp("in.Minv(%d)" % (n * b + a))

    # cols = [c0..c1] (column-split kernels): only those columns, written at LOCAL indices -- d/dq columns at n*(col-c0),
    # d/dqd columns at n*len(cols) + n*(col-c0); the kernel sink maps the two runs back (see _out_grad)
    lo_base = (lambda col: n * col) if cols is None else (lambda col: n * list(cols).index(col))
    hi_off = n * n if cols is None else n * len(cols)

    def emit_run(col, h, values):
        # lean cores: half-column runs in emission order -- through the sector-aligned stream (AlignedPieces), or as run k of the core
        # (n values) that goes to row offset tr.run_bases[k] (grid_out_runs)
        if stream is not None:
            stream.push(n * col + h * n * n, list(values))
            return
        k_run = len(tr.run_bases)
        tr.run_bases.append(n * col + h * n * n)
        for r in range(n):
            tr.out(n * k_run + r, values[r])

    def emit_column(col, dc):
        memo.clear()
        rows = sorted(dc)
        if kind != "fd" and want is not None:           # lean inverse-dynamics-gradient core: the column itself is the output
            for h in sorted(pass_halves.pop(0) if pass_halves else want[col]):
                emit_run(col, h, [dc[r][h] if r in dc else tr.zero() for r in range(n)])
            return
        if kind != "fd":
            for half in (0, 1):
                for r in range(n):
                    e = dc.get(r)
                    tr.out(half * hi_off + lo_base(col) + r, e[half] if e is not None else tr.zero())
            return
        if coop is not None:
            # every upper-triangle entry fetched once per column from the exchange region (4 multiply-adds per LDS read)
            dqd_half = saved_dqd.pop(col, None) or {k: dc[k][1] for k in rows}       # (computed ahead of the barriers when parked)
            halves = (pass_halves.pop(0) if pass_halves else want[col]) if want is not None else (0, 1)
            none = {k: tr.zero() for k in rows}
            if pairing["role"]:
                what = pairing["role"][pairing["at"]]
                pairing["at"] += 1
                h = halves[0]
                vec = {k: dc[k][0] for k in rows} if h == 0 else {k: dqd_half[k] for k in rows}
                if what == "first":
                    pairing["held"] = (col, h, vec)        # (its n or fewer values wait in registers for the partner's recursion)
                    return
                if what == "second":
                    (col_a, h_a, vec_a), pairing["held"] = pairing["held"], None
                    out_a, out_b = alg.sym_minv_times_column_pair(tr, spec, lambda r, k: minv_entry(r, k), vec_a, vec)
                    emit_run(col_a, h_a, out_a)
                    emit_run(col, h, out_b)
                    return
            if want is not None:
                if len(halves) == 2 and getattr(coop[1], "products_per_half", False):
                    # both halves of a column in one wave: ONE recursion, but the two products one after the other -- n accumulators
                    # alive instead of 2 n, for a second fetch of the Minv entries (2 multiply-adds per LDS read instead of 4)
                    lo, _ = alg.sym_minv_times_columns(tr, spec, lambda r, k: minv_entry(r, k), {k: dc[k][0] for k in rows}, none)
                    emit_run(col, 0, lo)
                    _, hi = alg.sym_minv_times_columns(tr, spec, lambda r, k: minv_entry(r, k), none, {k: dqd_half[k] for k in rows})
                    emit_run(col, 1, hi)
                    return
            lo, hi = alg.sym_minv_times_columns(tr, spec, lambda r, k: minv_entry(r, k),
                                                {k: dc[k][0] for k in rows} if 0 in halves else none,
                                                {k: dqd_half[k] for k in rows} if 1 in halves else none)
            if want is not None:
                for h in sorted(halves):
                    emit_run(col, h, (lo, hi)[h])
                return
            for r in range(n):
                tr.out(lo_base(col) + r, lo[r])
            for r in range(n):
                tr.out(hi_off + lo_base(col) + r, hi[r])
            return
        hi = []
        lo = []
        for r in range(n):
            m = [(k, minv_entry(r, k)) for k in rows]
            m = [(k, e) for (k, e) in m if e is not None]
            lo.append(-tr.dot([(e, dc[k][0]) for (k, e) in m]))
            if rollout is None:
                tr.out(lo_base(col) + r, lo[-1])
            hi.append(-tr.dot([(e, dc[k][1]) for (k, e) in m]))
        if rollout is not None:
            chunks.A_columns(col, lo, hi, dt, dt2)
            return
        for r in range(n):
            tr.out(hi_off + lo_base(col) + r, hi[r])

    keep = ([t.ref for row in Minv for t in row if t is not None] if Minv is not None else [])
    if rollout is not None:
        assert kind == "fd" and not use_qdd_minv and cols is None and not table
        keep = keep + [dt.ref, dt2.ref]
    order = cols
    if coop is not None:
        # phases 1 and 2 of the block (_coop_prologue).  A consumer fills its idle time with the d/dqd recursions of some of its
        # columns (CoopSlots.hoisted_columns): they need neither qdd nor Minv; the n values per column stay in registers until
        # the column's products after the second barrier.
        if lean:
            hoist = [c_ for c_ in role.hoist if c_ in cols and (want is None or 1 in want[c_])]
        else:
            hoist = slots.hoisted_columns(role, list(cols)) if role in ("consumer", "consumer_c") else []

        def pre_barrier():
            def capture(col, dc):
                memo.clear()
                saved_dqd[col] = {k: dc[k][1] for k in dc}
                for v in saved_dqd[col].values():
                    tr.anchor(v)
            alg.rnea_grad_columns(tr, spec, I, q, qd, trig, loader, capture, order=hoist, prefetch=0, xof=Xof, keep=keep,
                                  xof_back=Xof_back if lean else None, xa_first=lean)
        mark = tr.cse_mark()
        u = [tr.inp("in.u(%d)" % j) for j in range(n)] if kind == "fd" else None
        if lean and kind == "id":
            # register-lean inverse-dynamics-gradient core: only the block's input table (sin, cos, qd and, with use_qdd, the given
            # qdd in the qdd slots) and ONE barrier in front of the gradient half-columns
            qdd_in = [tr.inp("in.qdd(%d)" % j) for j in range(n)] if use_qdd else None
            for j in role.joints:
                if trig[j] is not None:
                    tr.xch_put(itab["s"][j], trig[j][0])
                    tr.xch_put(itab["c"][j], trig[j][1])
                else:
                    tr.xch_put(itab["q"][j], q[j])
                tr.xch_put(itab["qd"][j], qd[j])
                if use_qdd:
                    tr.xch_put(slots.qdd[j], qdd_in[j])
            tr.barrier()
            qdd = [tr.zero()] * n if use_qdd else None       # (placeholders: touch() reads the published qdd of every joint it visits)
        elif lean:
            # (mixed arithmetic: the Minv recursion -- alg.minv_backward_lean / minv_forward_lean -- and the qdd rows run in double
            #  INSIDE the waves; what crosses the block's LDS -- U, 1/D, the backward-pass entries, Minv, c, qdd -- crosses as float)
            # ---- phase 0: this wave's share of the block's input table, then B0 (everything below reads inputs through touch())
            for j in role.joints:
                if trig[j] is not None:
                    tr.xch_put(itab["s"][j], trig[j][0])
                    tr.xch_put(itab["c"][j], trig[j][1])
                else:
                    tr.xch_put(itab["q"][j], q[j])
                if not getattr(slots, "table_q_only", False):
                    tr.xch_put(itab["qd"][j], qd[j])
                    if "u" in itab:
                        tr.xch_put(itab["u"][j], u[j])
            tr.barrier()
            # ---- phase 1: backward pass of the Minv recursion (once per tree) | bias torques | parked d/dqd recursions, then B1
            def scratch(j, i):          # U_j (6) and 1/D_j of the backward pass: LDS words BELOW the exchange region -- the waves'
                return -(1 + 7 * j + i)   # staging regions, which nobody uses before the first output flush (after the last barrier)

            def where(kind_, j, i):
                return slots.minv[(j, i)] if kind_ == "M" else scratch(j, 6 if kind_ == "D" else i)
            partials = tr.mixed and getattr(slots, "partials", None) is not None      # (mixed arithmetic: lean_partial_layout)

            def publish_minv(kind_, j, i, val):
                tr.xch_put(where(kind_, j, i), val)
                if partials and kind_ == "M" and (j, i) in slots.root_lo and not isinstance(val.ref, float):
                    # a base joint's row is final here: its low part too, for the double partial sums of qdd
                    tr.xch_put(slots.root_lo[(j, i)], val - tr.cast(tr.cast(val, 0), 1))
            if role.minv_bwd:
                alg.minv_backward_lean(tr, spec, I, Xof_back, publish_minv, role.minv_bwd, cols=role.minv_bwd_cols)
                memo.clear()
            if role.c_roots:
                def publish_c(j, f):
                    c_j = f[spec.S_ind[j]] + qd[j] * spec.damping[j]
                    # (mixed arithmetic with double partial sums: c itself; every wave forms u - c in double from its own read of u)
                    tr.xch_put(slots.c[j], (u[j] - c_j) if (slots.lean_umc and not partials) else c_j)
                ftab["hook"] = publish_c
                for root in role.c_roots:
                    facc(root)
                ftab["hook"] = None
                memo.clear()
            if hoist:
                pre_barrier()
            tr.barrier()
            # ---- phase 2: forward pass of the Minv recursion for this wave's columns, then B2
            final = {}
            if role.minv_cols:
                def on_final(j, k, val):
                    if slots.minv.get((j, k)) is not None:
                        tr.xch_put(slots.minv[(j, k)], val)
                        final[(j, k)] = val
                if getattr(slots, "columns_from_chain", False):      # U, 1/D published only: the whole per-column recursion here
                    alg.minv_columns_lean(tr, spec, Xof_back, lambda kind_, j, i: tr.xch_get(where(kind_, j, i)), role.minv_cols, on_final)
                else:
                    alg.minv_forward_lean(tr, spec, Xof_back, lambda kind_, j, i: tr.xch_get(where(kind_, j, i)), role.minv_cols, on_final)
                memo.clear()
            if partials:
                # mixed arithmetic: this wave's share of qdd = Minv_sym (u - c) from the UNROUNDED entries of its columns (double),
                # B2, published as float pairs where U, 1/D and u - c lived, B2', then every row summed over the waves in double
                assert slots.lean_umc and getattr(role, "index", None) is not None or not role.minv_cols
                part, umc = {}, {}
                with tr.mixed_region():
                    for k in role.minv_cols:
                        for j in range(k + 1):
                            if (j, k) not in slots.minv:
                                continue
                            m = final.get((j, k))
                            if m is None:           # (a base joint's row: final since the backward pass, hi + lo)
                                m = tr.cast(tr.xch_get(slots.minv[(j, k)]), 1) + tr.xch_get(slots.root_lo[(j, k)])
                            for (row, col_) in ((j, k), (k, j)) if j != k else ((j, k),):
                                if col_ not in umc:
                                    umc[col_] = tr.cast(u[col_], 1) - tr.xch_get(slots.c[col_])
                                part[row] = tr.fma(m, umc[col_], part[row]) if row in part else m * umc[col_]
                tr.barrier()
                if role.minv_cols:
                    first, rows_w = slots.partials[role.index]
                    for i_, r in enumerate(rows_w):
                        p_ = part.get(r, tr.zero())
                        hi_ = tr.cast(p_, 0)
                        tr.xch_put(-(1 + 2 * (first + i_)), hi_)
                        tr.xch_put(-(2 + 2 * (first + i_)), (p_ - tr.cast(hi_, 1)) if not isinstance(p_.ref, float) else tr.zero())
                tr.barrier()
                for r in role.qdd_rows:
                    with tr.mixed_region():
                        total = None
                        for w_, (first, rows_w) in sorted(slots.partials.items()):
                            if r in rows_w:
                                i_ = rows_w.index(r)
                                pair = tr.cast(tr.xch_get(-(1 + 2 * (first + i_))), 1) + tr.xch_get(-(2 + 2 * (first + i_)))
                                total = pair if total is None else total + pair
                    tr.xch_put(slots.qdd[r], total)
            else:
                tr.barrier()
            if getattr(role, "out_minv", False) and role.minv_cols:
                # Minv kernel (lean_plan_minv): this wave's columns, upper triangle, column by column (n values at row offset n*k of the
                # n x n output; rows below the diagonal and rows of other trees are zero) -- AFTER B2: a flush goes through the wave's
                # staging region, where U and 1/D are parked until every wave has finished its forward pass
                for k in role.minv_cols:
                    col_vals = [tr.xch_get(slots.minv[(r, k)]) if (r <= k and (r, k) in slots.minv) else tr.zero() for r in range(n)]
                    for at in range(0, n, AlignedPieces.MAX_PIECE):
                        chunk = col_vals[at:at + AlignedPieces.MAX_PIECE]
                        for pos, v_ in enumerate(chunk):
                            tr.out("piece:%d:%d" % (len(chunk), pos), v_)
                        tr.out("flush:%d:%d" % (len(chunk), n * k + at), 0.0)
            # ---- phase 3: rows of qdd = Minv_sym (u - c) from the published Minv and c, then B3
            if role.qdd_rows and not partials:
                umc = {}
                for r in role.qdd_rows:
                    terms = []
                    for k in range(n):
                        m = slots.entry(tr, r, k)
                        if m is None:
                            continue
                        if k not in umc:
                            umc[k] = tr.xch_get(slots.c[k]) if slots.lean_umc else tr.xch_get(itab["u"][k]) - tr.xch_get(slots.c[k])
                        terms.append((m, umc[k]))
                    with tr.mixed_region():
                        row = tr.dot(terms)
                    tr.xch_put(slots.qdd[r], row)
            tr.barrier()
            qdd = [tr.zero()] * n       # (placeholders: touch() reads the published qdd of every joint it visits)
            if getattr(role, "out_qdd", False):
                # forward-dynamics kernel (lean_plan_fd): the block's result is qdd itself -- this wave writes the n values per
                # configuration (row length n), in pieces of at most 32
                vals = [tr.xch_get(slots.qdd[r]) for r in range(n)]
                for at in range(0, n, AlignedPieces.MAX_PIECE):
                    chunk = vals[at:at + AlignedPieces.MAX_PIECE]
                    for pos, v_ in enumerate(chunk):
                        tr.out("piece:%d:%d" % (len(chunk), pos), v_)
                    tr.out("flush:%d:%d" % (len(chunk), at), 0.0)
        else:
            X = alg.build_X(tr, spec, q, trig)
            qdd = list(_coop_prologue(tr, spec, slots, role, X, I, list(qd), u, g, demand_order=False, pre_barrier=pre_barrier if hoist else None))
        tr.fence()
        kept_trig = [] if itab is not None else [t.ref for pair in trig if pair is not None for t in pair]
        tr.cse_release(mark, keep=kept_trig + [v.ref for d in saved_dqd.values() for v in d.values() if not isinstance(v.ref, float)])
        memo.clear()
        # the parked columns first (their registers are freed early), and within both runs the deepest first: a column's accumulated
        # force then finds its children's in the f table (DFS pre-order ids: children have larger ids)
        order = sorted(hoist, reverse=True) + sorted((c for c in cols if c not in hoist), reverse=True)
        if lean and getattr(slots, "separate_halves", False):
            # a column whose two halves are both this wave's and not parked: TWO passes over the column (the d/dq recursion, then the
            # d/dqd recursion) instead of one carrying both -- half the path state alive, for the walk over the joints done twice
            passes = []
            for c_ in order:
                if len(want[c_]) == 2 and c_ not in hoist:
                    passes += [(c_, (0,)), (c_, (1,))]
                else:
                    passes.append((c_, tuple(sorted(want[c_]))))
            order = [c_ for (c_, _) in passes]
            pass_halves.extend(h_ for (_, h_) in passes)
        if lean and getattr(slots, "aligned_flush", False) and want is not None:
            # one pass per half-column, the d/dqd columns in ascending order (the parked ones first: lean_plan(order="runs") parks the
            # first columns of the run), then the d/dq columns in ascending order: neighbours of the output row follow each other
            hi_cols = sorted(c_ for c_ in want if 1 in want[c_])
            hi_cols = [c_ for c_ in hi_cols if c_ in hoist] + [c_ for c_ in hi_cols if c_ not in hoist]
            passes = [(c_, (1,)) for c_ in hi_cols] + [(c_, (0,)) for c_ in sorted((c_ for c_ in want if 0 in want[c_]),
                                                                                      reverse=getattr(slots, "chain_f", False))]
            order = [c_ for (c_, _) in passes]
            del pass_halves[:]
            pass_halves.extend(h_ for (_, h_) in passes)
            if getattr(slots, "pair_products", False) and kind == "fd":
                # consecutive passes of the same half and the same base-rooted tree share ONE product (alg.sym_minv_times_column_pair)
                root = lambda j: j if spec.parent[j] == -1 else root(spec.parent[j])
                role_of, i = [None] * len(passes), 0
                while i + 1 < len(passes):
                    (ca, ha), (cb, hb) = passes[i], passes[i + 1]
                    if ha == hb and root(ca) == root(cb):
                        role_of[i], role_of[i + 1] = "first", "second"
                        i += 2
                    else:
                        i += 1
                pairing["role"] = role_of
            seq = []                        # row offsets of the half-columns in the order they will be emitted
            halves_of = list(pass_halves) if pass_halves else [tuple(sorted(want[c_])) for c_ in order]
            for c_, hs in zip(order, halves_of):
                seq += [n * c_ + h_ * n * n for h_ in hs]
            stream = AlignedPieces(tr, n, 2 * n * n, seq)
        ftab["on"] = bool(slots.f_table)
    alg.rnea_grad_columns(tr, spec, I, q, qd, trig, loader, emit_column, order=order, prefetch=0, xof=Xof, keep=keep,
                          xof_back=Xof_back if lean else None, xa_first=lean)
    return tr
