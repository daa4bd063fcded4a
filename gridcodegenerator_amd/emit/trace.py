"""Scalar straight-line tracer: the generation-time specialiser behind every emitted ``_inner``.

The reference emits loops over dense 6x6 products whose bounds and offsets are literals
(``algorithms/_inverse_dynamics.py:151-167``: ``dot_prod<T,6,6,1>`` over full rows even though half
of X is structurally zero).  Here the algorithm is *run once in Python* on symbolic scalars:

* a value is either a Python float (known at generation time) or a signed reference to an IR node;
* ``0*x``, ``1*x``, ``x+0`` and constant arithmetic fold away, so structural zeros / +-1 entries of
  X_j(q) and I_j never reach the device code;
* nodes are hash-consed (common sub-expressions such as the ``X_j v_parent`` shared by the two RNEA
  passes of the forward-dynamics gradient are emitted once);
* dead nodes are dropped at emission (only what reaches an output survives).

The result is one basic block of ``fma``/``mul``/``add`` on named temporaries in the compute type
``C`` -- one wavefront lane executes it for one configuration; there is no cross-lane traffic and no
barrier inside an ``_inner``.
"""
import math


class V:
    """A traced scalar: ``ref`` is a Python float or a signed node index (+k / -k, k >= 1)."""
    __slots__ = ("tr", "ref")

    def __init__(self, tr, ref):
        self.tr = tr
        self.ref = ref

    # --- helpers -------------------------------------------------------------------------------
    def is_const(self):
        return isinstance(self.ref, float)

    def is_zero(self):
        return isinstance(self.ref, float) and self.ref == 0.0

    def const(self):
        return self.ref

    def _lift(self, o):
        if isinstance(o, V):
            return o
        return V(self.tr, float(o))

    # --- arithmetic ------------------------------------------------------------------------------
    def __neg__(self):
        return V(self.tr, -self.ref)

    def __add__(self, o):
        if isinstance(o, P):
            return o + self
        return self.tr.add(self, self._lift(o))

    __radd__ = __add__

    def __sub__(self, o):
        if isinstance(o, P):
            return (-o) + self
        return self.tr.add(self, -self._lift(o))

    def __rsub__(self, o):
        return self.tr.add(self._lift(o), -self)

    def __mul__(self, o):
        if isinstance(o, P):
            return o * self
        return self.tr.mul(self, self._lift(o))

    __rmul__ = __mul__

    def __repr__(self):
        return "V(%r)" % (self.ref,)


class P:
    """A pair of traced scalars that undergo the same operations (here: the d/dq and d/dqd versions of a gradient
    quantity).  Arithmetic on pairs is emitted as ONE packed instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
    on a 64-bit register pair) whenever both halves need a real operation; a lone wavefront issues a packed
    instruction at the same cadence as a scalar one (tools/ubench/issue_rate.hip), so this halves the instruction
    count of the gradient recursions.  Halves that fold (zeros, +-1, constants) stay scalar."""
    __slots__ = ("tr", "lo", "hi")

    def __init__(self, tr, lo, hi):
        self.tr = tr
        self.lo = lo if isinstance(lo, V) else V(tr, float(lo))
        self.hi = hi if isinstance(hi, V) else V(tr, float(hi))

    def is_zero(self):
        return self.lo.is_zero() and self.hi.is_zero()

    def __neg__(self):
        return P(self.tr, -self.lo, -self.hi)

    def __add__(self, o):
        return self.tr.pair_op("add", self, o, None)

    __radd__ = __add__

    def __sub__(self, o):
        return self.tr.pair_op("add", self, -o if isinstance(o, (P, V)) else -float(o), None)

    def __rsub__(self, o):
        return self.tr.pair_op("add", -self, o, None)

    def __mul__(self, o):
        return self.tr.pair_op("mul", self, o, None)

    __rmul__ = __mul__

    def __repr__(self):
        return "P(%r, %r)" % (self.lo.ref, self.hi.ref)


class Tracer:
    def __init__(self):
        self.nodes = [None]      # 1-based; each node: (op, a, b, c) with operand refs / payload
        self.ntype = [0]         # per node: 0 = compute type C (float), 1 = high-precision type D (double)
        self.hp = False          # arithmetic created while True is typed D (mixed precision, see high())
        self.cse = {}
        self.outputs = []        # (dst_expr, ref)
        self.out_pos = []        # node count when each output was recorded
        self.fences = []         # node counts at which a scheduling fence was requested
        self.comments = {}       # node index -> comment emitted before it

    # --- leaf constructors ---------------------------------------------------------------------
    def const(self, x):
        return V(self, float(x))

    def zero(self):
        return V(self, 0.0)

    def inp(self, expr):
        """Load of an input element (a C expression such as ``s_q[3]``)."""
        return V(self, self._node(("in", expr, None, None)))

    def _type_of_new(self, key):
        op = key[0]
        if op == "in":
            return 0                                  # inputs arrive in the storage type
        if op in ("lnd", "lo", "hi", "bc"):
            return self.ntype[key[1]]                 # same value, same type
        if op == "cvt":
            return key[2]                             # explicit conversion to C (0) or D (1)
        return 1 if self.hp else 0

    def _node(self, key):
        ty = self._type_of_new(key)
        idx = self.cse.get((key, ty))
        if idx is None:
            self.nodes.append(key)
            self.ntype.append(ty)
            idx = len(self.nodes) - 1
            self.cse[(key, ty)] = idx
        return idx

    def high(self):
        """Context manager: arithmetic traced inside is carried out in the high-precision type D (double).  Operands of
        the other type are converted at the use (widening is exact; narrowing rounds once)."""
        tr = self

        class _High:
            def __enter__(self_inner):
                self_inner.prev = tr.hp
                tr.hp = True

            def __exit__(self_inner, *exc):
                tr.hp = self_inner.prev
        return _High()

    mixed = False           # class-wide switch (GRiDCodeGenerator(precision="mixed") sets it for a build)

    def mixed_region(self):
        """high() when the build is mixed-precision, otherwise a no-op: wraps the cond(M)-amplified parts of the forward
        dynamics (the Minv recursion and qdd = Minv (u - c)), see DESIGN.md section 4."""
        if self.mixed:
            return self.high()
        import contextlib
        return contextlib.nullcontext()

    def cast(self, a, hp):
        """The same value converted ONCE to the compute type C (hp = 0, rounds) or D (hp = 1, exact): a node of its own, so
        every later use shares the conversion and the original may die."""
        if isinstance(a.ref, float) or self.ntype[abs(a.ref)] == hp:
            return a
        sign = 1 if a.ref > 0 else -1
        return V(self, sign * self._node(("cvt", abs(a.ref), hp, None)))

    def low(self):
        """Context manager: the opposite of high() (compute type C inside)."""
        tr = self

        class _Low:
            def __enter__(self_inner):
                self_inner.prev = tr.hp
                tr.hp = False

            def __exit__(self_inner, *exc):
                tr.hp = self_inner.prev
        return _Low()

    def comment(self, text):
        self.comments.setdefault(len(self.nodes), []).append(text)

    # --- operations --------------------------------------------------------------------------------
    def mul(self, a, b):
        ra, rb = a.ref, b.ref
        if isinstance(ra, float) and isinstance(rb, float):
            return V(self, ra * rb)
        if isinstance(ra, float):
            ra, rb = rb, ra
        # now ra is a node ref; rb may be a float
        if isinstance(rb, float):
            if rb == 0.0:
                return V(self, 0.0)
            if rb == 1.0:
                return V(self, ra)
            if rb == -1.0:
                return V(self, -ra)
            sign = (1 if ra > 0 else -1) * (1 if rb > 0 else -1)
            return V(self, sign * self._node(("mul", abs(ra), abs(rb), None)))
        sign = (1 if ra > 0 else -1) * (1 if rb > 0 else -1)
        x, y = sorted((abs(ra), abs(rb)))
        return V(self, sign * self._node(("mul", x, y, None)))

    def add(self, a, b):
        ra, rb = a.ref, b.ref
        if isinstance(ra, float) and isinstance(rb, float):
            return V(self, ra + rb)
        if isinstance(ra, float):
            ra, rb = rb, ra
        if isinstance(rb, float):
            if rb == 0.0:
                return V(self, ra)
            fused = self._fuse_mul_add(ra, rb)
            if fused is not None:
                return fused
            if ra < 0:  # -(x) + c == -(x - c)
                return V(self, -self._node(("add", -ra, -rb, None)))
            return V(self, self._node(("add", ra, rb, None)))
        if ra == -rb:
            return V(self, 0.0)
        if abs(ra) > abs(rb):
            ra, rb = rb, ra
        # structural fusion: (x*y) + c becomes fma(x, y, c) whenever an operand is a product.  Decided from the
        # expression alone (never from use counts), so every variant of a trace rounds identically; the build
        # passes -ffp-contract=off so the compiler adds no fusions of its own.
        fused = self._fuse_mul_add(rb, ra)
        if fused is None:
            fused = self._fuse_mul_add(ra, rb)
        if fused is not None:
            return fused
        if ra < 0:
            return V(self, -self._node(("add", -ra, -rb, None)))
        return V(self, self._node(("add", ra, rb, None)))

    def _fuse_mul_add(self, rm, rc):
        """rm: signed node ref; if it is a product node return fma(x, y, rc) (sign-correct), else None."""
        if isinstance(rm, float):
            return None
        op, x, y, _ = self.nodes[abs(rm)]
        if op != "mul":
            return None
        vx = V(self, x if rm > 0 else -x)
        vy = V(self, y)
        return self.fma(vx, vy, V(self, rc))

    # --- paired (2-wide) operations ------------------------------------------------------------------
    use_packed = False      # class-wide switch (GRiDCodeGenerator(packed=...) sets it for a build); off: what ships

    def _halves(self, x):
        if isinstance(x, P):
            return x.lo, x.hi
        if isinstance(x, V):
            return x, x
        v = V(self, float(x))
        return v, v

    def pair_op(self, kind, a, b, c):
        """kind in {"fma", "mul", "add"} on operands that are P, V or numbers.  Both halves are first formed with
        the scalar rules (all folding applies); if BOTH halves turned into brand-new arithmetic nodes, those two nodes
        are withdrawn and replaced by one packed node."""
        aL, aH = self._halves(a)
        bL, bH = self._halves(b)
        cL, cH = self._halves(c) if c is not None else (None, None)
        mark = len(self.nodes)

        def scalar(x, y, z):
            if kind == "fma":
                return self.fma(x, y, z)
            if kind == "mul":
                return self.mul(x, y)
            return self.add(x, y)

        rL = scalar(aL, bL, cL)
        n_after_lo = len(self.nodes)
        rH = scalar(aH, bH, cH)
        fresh_lo = (not isinstance(rL.ref, float)) and abs(rL.ref) >= mark and abs(rL.ref) == n_after_lo - 1 and n_after_lo == mark + 1
        fresh_hi = (not isinstance(rH.ref, float)) and abs(rH.ref) == len(self.nodes) - 1 and len(self.nodes) == n_after_lo + 1
        if not (self.use_packed and fresh_lo and fresh_hi):
            return P(self, rL, rH)
        # gfx950's VOP3P encoding has no literal operand: a model constant inside a packed instruction has to come from an SGPR
        # (s_mov_b32 with a literal costs a lone wave 8 cycles; tools/ubench/packed_cost.hip: 6.4 cycles per result against 5.5
        # for two scalar v_fmaak_f32).  Operations with a constant operand therefore stay two scalar instructions on the halves;
        # only (pair x per-lane value) and (pair + pair) are packed (3.0 cycles per result).
        operands = [aL, aH, bL, bH] + ([cL, cH] if cL is not None else [])
        if any(isinstance(o.ref, float) and o.ref != 0.0 for o in operands):
            return P(self, rL, rH)
        kL, kH = self.nodes[abs(rL.ref)][0], self.nodes[abs(rH.ref)][0]
        if kL not in ("fma", "mul", "add") or kH not in ("fma", "mul", "add"):
            return P(self, rL, rH)
        # withdraw the two scalar nodes (they are the last two appended) and emit one packed node instead
        for _ in range(2):
            key = self.nodes.pop()
            del self.cse[(key, self.ntype.pop())]
        if kind == "fma":
            key = ("pkfma", (aL.ref, aH.ref), (bL.ref, bH.ref), (cL.ref, cH.ref))
        elif kind == "mul":
            key = ("pkmul", (aL.ref, aH.ref), (bL.ref, bH.ref), None)
        else:
            key = ("pkadd", (aL.ref, aH.ref), (bL.ref, bH.ref), None)
        k = self._node(key)
        lo = self._node(("lo", k, None, None))
        hi = self._node(("hi", k, None, None))
        return P(self, V(self, lo), V(self, hi))

    def fma(self, a, b, c):
        """a*b + c with folding."""
        if isinstance(a, P) or isinstance(b, P) or isinstance(c, P):
            return self.pair_op("fma", a, b, c)
        ra, rb, rc = a.ref, b.ref, c.ref
        if isinstance(rc, float) and rc == 0.0:
            return self.mul(a, b)
        if isinstance(ra, float) and isinstance(rb, float):
            return self.add(V(self, ra * rb), c)
        if isinstance(ra, float):
            ra, rb = rb, ra
        if isinstance(rb, float):
            if rb == 0.0:
                return c
            if rb == 1.0:
                return self.add(V(self, ra), c)
            if rb == -1.0:
                return self.add(V(self, -ra), c)
            sign = (1 if ra > 0 else -1) * (1 if rb > 0 else -1)
            x, y = abs(ra), abs(rb)
        else:
            sign = (1 if ra > 0 else -1) * (1 if rb > 0 else -1)
            x, y = sorted((abs(ra), abs(rb)))
        # sign*(x*y) + rc ; canonical form keeps the product positive
        if sign > 0:
            return V(self, self._node(("fma", x, y, rc)))
        return V(self, -self._node(("fma", x, y, -rc)))

    dot_ways = 1            # class-wide: > 1 splits long dot products over that many independent accumulators

    def dot(self, pairs, init=None):
        """sum_i a_i*b_i (+ init) as an fma chain; zero terms vanish.  With dot_ways = w > 1 a product of >= 2w non-zero
        terms is accumulated in w interleaved chains that are added at the end (shorter dependency chain, one more add per
        extra chain, different rounding)."""
        lifted = []
        for (a, b) in pairs:
            a = a if isinstance(a, (V, P)) else V(self, float(a))
            b = b if isinstance(b, (V, P)) else V(self, float(b))
            if not (a.is_zero() or b.is_zero()):
                lifted.append((a, b))
        w = self.dot_ways
        if w > 1 and len(lifted) >= 2 * w:
            accs = [init if (init is not None and k == 0) else V(self, 0.0) for k in range(w)]
            for i, (a, b) in enumerate(lifted):
                accs[i % w] = self.fma(a, b, accs[i % w])
            acc = accs[0]
            for k in range(1, w):
                acc = acc + accs[k]
            return acc
        acc = init if init is not None else V(self, 0.0)
        for (a, b) in lifted:
            acc = self.fma(a, b, acc)
        return acc

    def rcp(self, a):
        if isinstance(a.ref, float):
            return V(self, 1.0 / a.ref)
        sign = 1 if a.ref > 0 else -1
        return V(self, sign * self._node(("rcp", abs(a.ref), None, None)))

    def sin(self, a):
        if isinstance(a.ref, float):
            return V(self, math.sin(a.ref))
        sign = 1 if a.ref > 0 else -1
        return V(self, sign * self._node(("sin", abs(a.ref), None, None)))

    def cos(self, a):
        if isinstance(a.ref, float):
            return V(self, math.cos(a.ref))
        return V(self, self._node(("cos", abs(a.ref), None, None)))

    def out(self, dst_expr, val):
        val = val if isinstance(val, V) else V(self, float(val))
        self.outputs.append((dst_expr, val.ref))
        self.out_pos.append(len(self.nodes))       # creation-order emission places the store here

    def tab_put(self, slot, val):
        """Park a value in the lane-private table (LDS in the kernels, a local array in the _device functions)."""
        self.out("tab:%d" % slot, val)

    def tab_get(self, slot):
        """Re-load a parked value: a fresh node per request, so its live range starts here."""
        self._tab_serial = getattr(self, "_tab_serial", 0) + 1
        return self.inp("in.tab_get(%d)/*%d*/" % (slot, self._tab_serial))

    # --- tile-cooperative kernels: the waves of a block exchange per-configuration values through LDS ---------------------
    def xch_put(self, slot, val):
        """Publish a value to the block's exchange region (LDS): read by the same lane index of the OTHER waves of the block
        after the barrier."""
        self.out("xch:%d" % slot, val)

    def xch_get(self, slot):
        """Read a published value: a fresh load per request (short live range, like tab_get)."""
        self._xch_serial = getattr(self, "_xch_serial", 0) + 1
        return self.inp("in.xch_get(%d)/*%d*/" % (slot, self._xch_serial))

    def barrier(self):
        """Block barrier at this point of the core (every wave of the block executes exactly the same number of them)."""
        self.out("barrier", 0.0)

    def anchor(self, val):
        """Force `val` to be computed before this point (work that should overlap with another wave's producer phase)."""
        self.out("anchor", val)

    # --- wave-per-configuration kernels: the lanes of ONE wavefront share a configuration (emit/wave.py) ------------------------
    def bcast(self, a, lane):
        """The value lane `lane` (a generation-time constant) holds, in every lane: a wave-uniform value (v_readlane_b32)."""
        if isinstance(a.ref, float):
            return a
        assert self.ntype[abs(a.ref)] == 0, "broadcasts are done in the compute type C"
        sign = 1 if a.ref > 0 else -1
        return V(self, sign * self._node(("bc", abs(a.ref), int(lane), None)))

    def utab_put(self, slot, val):
        """Park a WAVE-UNIFORM value in the wave's LDS table (one word per slot: every lane writes the same value)."""
        self.out("utab:%d" % slot, val)

    def utab_get(self, slot):
        """Re-load a parked uniform value (LDS broadcast read): a fresh node per request, so its live range starts here."""
        self._utab_serial = getattr(self, "_utab_serial", 0) + 1
        return self.inp("in.utab_get(%d)/*%d*/" % (slot, self._utab_serial))

    def m_put(self, row, val):
        """Lane k publishes entry [row][k] of a matrix whose column k it owns (Minv) to the wave's LDS."""
        self.out("mput:%d" % row, val)

    def m_get(self, row, col):
        """Entry [row][col] of the published matrix as a wave-uniform value (LDS broadcast read); fresh node per request."""
        self._m_serial = getattr(self, "_m_serial", 0) + 1
        return self.inp("in.m_get(%d,%d)/*%d*/" % (row, col, self._m_serial))

    def wave_sync(self):
        """Wave-local LDS ordering point (s_waitcnt lgkmcnt(0) + compiler fences): lanes read what OTHER lanes of the wave wrote."""
        self.out("wsync", 0.0)

    def launder(self, a):
        """Same value, but opaque to the compiler from here on (an empty asm with the register as in/out operand): a later
        expression over the laundered value is NOT a common subexpression of the same expression over the original, so
        hipcc cannot undo a deliberate recomputation by keeping the first result alive (and spilling it)."""
        if isinstance(a.ref, float):
            return a
        sign = 1 if a.ref > 0 else -1
        self._launder_serial = getattr(self, "_launder_serial", 0) + 1
        return V(self, sign * self._node(("lnd", abs(a.ref), self._launder_serial, None)))

    def fence(self):
        """Scheduling fence at this point of the trace (creation-order emission only)."""
        self.fences.append(len(self.nodes))

    def cse_mark(self):
        return len(self.nodes)

    def cse_release(self, mark, keep=()):
        """Forget the common-subexpression entries of every node created since `mark` (except `keep` refs): a later
        identical expression is then RECOMPUTED (or, for an input, re-loaded) instead of extending the old value's life.
        This is how the explicit schedules rematerialise X_j(q) entries and reload workspace values per column."""
        keep = set(abs(r) for r in keep if not isinstance(r, float))
        for k in range(mark, len(self.nodes)):
            if k not in keep and self.cse.get((self.nodes[k], self.ntype[k])) == k:
                del self.cse[(self.nodes[k], self.ntype[k])]

    # --- analysis / emission ---------------------------------------------------------------------
    def _deps(self, k):
        op, a, b, c = self.nodes[k]
        if op == "in":
            return ()
        if op in ("lo", "hi", "lnd", "cvt", "bc"):
            return (a,)
        if op.startswith("pk"):
            return tuple(abs(r) for pair in (a, b, c) if pair is not None for r in pair if not isinstance(r, float))
        return tuple(abs(r) for r in (a, b, c) if r is not None and not isinstance(r, float))

    def live_nodes(self):
        live = [False] * len(self.nodes)
        stack = [abs(r) for (_, r) in self.outputs if not isinstance(r, float)]
        while stack:
            k = stack.pop()
            if live[k]:
                continue
            live[k] = True
            for d in self._deps(k):
                if not live[d]:
                    stack.append(d)
        return live

    def op_counts(self):
        live = self.live_nodes()
        counts = {}
        for k in range(1, len(self.nodes)):
            if live[k]:
                name = self.nodes[k][0] + (".d" if self.ntype[k] else "")      # ".d": carried out in double
                counts[name] = counts.get(name, 0) + 1
        return counts

    def flops(self):
        c = self.op_counts()
        return (2 * (c.get("fma", 0) + c.get("fma.d", 0)) + c.get("mul", 0) + c.get("add", 0) + c.get("mul.d", 0) + c.get("add.d", 0)
                + 4 * c.get("pkfma", 0) + 2 * c.get("pkmul", 0) + 2 * c.get("pkadd", 0))

    def arith_instructions(self):
        """Arithmetic instructions one lane issues (a packed op is one instruction)."""
        c = self.op_counts()
        return sum(c.get(k, 0) for k in ("fma", "mul", "add", "fma.d", "mul.d", "add.d", "pkfma", "pkmul", "pkadd"))

    @staticmethod
    def _lit(x, hp=0):
        ty = "D" if hp else "C"
        if x == int(x) and abs(x) < 1e9:
            return "(%s)%d" % (ty, int(x))
        return "(%s)%s" % (ty, repr(float(x)))

    def _opnd(self, r, hp=0):
        """Operand text in the type of the consuming operation (hp: D, else C); converts where the node's type differs."""
        if isinstance(r, float):
            return self._lit(r, hp)
        op = self.nodes[abs(r)][0]
        if op in ("lo", "hi"):
            name = "t%d.%s" % (self.nodes[abs(r)][1], "x" if op == "lo" else "y")
        else:
            name = "t%d" % abs(r)
        if self.ntype[abs(r)] != hp:
            name = "(%s)%s" % ("D" if hp else "C", name)
        return name if r > 0 else "-" + name

    def _pair_opnd(self, pair):
        l, h = pair
        if not isinstance(l, float) and not isinstance(h, float) and (l > 0) == (h > 0):
            nl, nh = self.nodes[abs(l)], self.nodes[abs(h)]
            if nl[0] == "lo" and nh[0] == "hi" and nl[1] == nh[1]:
                return ("t%d" % nl[1]) if l > 0 else ("-t%d" % nl[1])
        return "C2{%s, %s}" % (self._opnd(l), self._opnd(h))

    def emit(self, indent="    ", order="demand", store=None, after_store=None, fence_every=0, fence_stmt="GRID_SCHED_FENCE();", read_ahead=0):
        """C++ statements (compute type ``C``, storage type ``T``) for all live nodes + output stores.

        order="demand": outputs are visited in order and each pulls in (post-order) whatever it still
        needs, so a value is computed close to its first use -- this keeps live ranges short in the
        long matrix-product tails.  order="creation": nodes in trace order, all stores at the end.
        store(dst, value_expr) -> statement; after_store(i) -> optional extra statement after output i.
        fence_every=N > 0 inserts ``GRID_SCHED_FENCE();`` every N statements: the emitted order already keeps
        live ranges short, and without fences hipcc's machine scheduler re-orders the single giant basic block
        for ILP and doubles the register pressure (measured: iiwa-7 FD gradient 472 -> 257 registers).
        """
        live = self.live_nodes()
        if store is None:
            store = lambda dst, val: "%s = (T)(%s);" % (dst, val)
        trig_args = {}
        for k in range(1, len(self.nodes)):
            if live[k] and self.nodes[k][0] in ("sin", "cos"):
                trig_args.setdefault((self.nodes[k][1], self.ntype[k]), {})[self.nodes[k][0]] = k
        emitted = [False] * len(self.nodes)
        lines = []
        count = [0]

        def emit_node(k):
            op, a, b, c = self.nodes[k]
            hp = self.ntype[k]
            ty = "D" if hp else "C"
            o = lambda r: self._opnd(r, hp)
            emitted[k] = True
            count[0] += 1
            if fence_every and count[0] % fence_every == 0:
                lines.append(indent + "GRID_SCHED_FENCE();")
            if op == "in":
                lines.append("%sconst C t%d = (C)%s;" % (indent, k, a))
            elif op == "mul":
                lines.append("%sconst %s t%d = %s * %s;" % (indent, ty, k, o(a), o(b)))
            elif op == "add":
                if not isinstance(b, float) and b < 0:
                    lines.append("%sconst %s t%d = %s - %s;" % (indent, ty, k, o(a), o(-b)))
                elif isinstance(b, float) and b < 0:
                    lines.append("%sconst %s t%d = %s - %s;" % (indent, ty, k, o(a), self._lit(-b, hp)))
                else:
                    lines.append("%sconst %s t%d = %s + %s;" % (indent, ty, k, o(a), o(b)))
            elif op == "fma":
                lines.append("%sconst %s t%d = grid_fma(%s, %s, %s);" % (indent, ty, k, o(a), o(b), o(c)))
            elif op == "rcp":
                lines.append("%sconst %s t%d = grid_rcp(%s);" % (indent, ty, k, o(a)))
            elif op == "lnd":
                count[0] -= 1       # no instruction when the source dies here (the usual case)
                lines.append("%s%s t%d = %s; GRID_LAUNDER(t%d);" % (indent, ty, k, o(a), k))
            elif op in ("lo", "hi"):
                count[0] -= 1       # a register half of a packed value: no instruction
            elif op == "cvt":
                lines.append("%sconst %s t%d = (%s)t%d;" % (indent, ty, k, ty, a))
            elif op == "bc":
                lines.append("%sconst %s t%d = grid_lanes::bcast(t%d, %d);" % (indent, ty, k, a, b))
            elif op == "pkfma":
                lines.append("%sconst C2 t%d = grid_pk_fma(%s, %s, %s);" % (indent, k, self._pair_opnd(a), self._pair_opnd(b), self._pair_opnd(c)))
            elif op == "pkmul":
                lines.append("%sconst C2 t%d = %s * %s;" % (indent, k, self._pair_opnd(a), self._pair_opnd(b)))
            elif op == "pkadd":
                lines.append("%sconst C2 t%d = %s + %s;" % (indent, k, self._pair_opnd(a), self._pair_opnd(b)))
            elif op in ("sin", "cos"):
                pair = trig_args[(a, hp)]
                if "sin" in pair and "cos" in pair:
                    emitted[pair["sin"]] = True
                    emitted[pair["cos"]] = True
                    lines.append("%s%s t%d, t%d; grid_sincos(%s, &t%d, &t%d);" % (
                        indent, ty, pair["sin"], pair["cos"], o(a), pair["sin"], pair["cos"]))
                else:
                    lines.append("%sconst %s t%d = grid_%s(%s);" % (indent, ty, k, op, o(a)))
            else:
                raise AssertionError(op)

        deps = self._deps

        if order == "creation":
            # nodes in trace order; every store where the algorithm recorded it; explicit fences honoured
            events = sorted([(pos, 0, i) for i, pos in enumerate(self.out_pos)] + [(pos, 1, -1) for pos in self.fences])
            ev = 0
            # read_ahead = L > 0: reads of the block's LDS regions (exchange slots, parked values) are issued L live nodes AHEAD of the
            # place the algorithm asked for them -- an LDS round trip is 64-128 cycles, a lone or older wave issues an instruction per
            # ~4, and the scheduling fences keep hipcc from hoisting a read out of its region by itself -- but never above a block
            # barrier nor above this wave's own write to the same slot.  Same values, earlier requests.
            early = {}
            if read_ahead > 0:
                live_idx = [k for k in range(1, len(self.nodes)) if live[k]]
                rank = {k: i for i, k in enumerate(live_idx)}
                floor_all = 1                      # after the latest barrier so far
                last_put = {}                      # slot -> position of this wave's latest write to it
                outs_sorted = sorted(zip(self.out_pos, range(len(self.outputs))))
                oi = 0
                for k in live_idx:
                    while oi < len(outs_sorted) and outs_sorted[oi][0] <= k:
                        pos, i = outs_sorted[oi]
                        dst = self.outputs[i][0]
                        if dst == "barrier":
                            floor_all = pos
                        elif isinstance(dst, str) and (dst.startswith("tab:") or dst.startswith("xch:")):
                            last_put[dst.split(":")[1]] = pos
                        oi += 1
                    op, a = self.nodes[k][0], self.nodes[k][1]
                    if op == "in" and isinstance(a, str) and (a.startswith("in.xch_get(") or a.startswith("in.tab_get(")):
                        slot = a[a.index("(") + 1:a.index(")")]
                        lo = max(floor_all, last_put.get(slot, 1))
                        target = live_idx[max(0, rank[k] - read_ahead)]
                        target = max(target, lo)
                        if target < k:
                            early.setdefault(target, []).append(k)
            for k in range(1, len(self.nodes) + 1):
                while ev < len(events) and events[ev][0] <= k:
                    _, kind, i = events[ev]
                    ev += 1
                    if kind == 1:
                        lines.append(indent + fence_stmt)
                        continue
                    dst, r = self.outputs[i]
                    lines.append(indent + store(dst, self._opnd(r, self._store_type(r))))
                    if after_store is not None:
                        extra = after_store(i)
                        if extra:
                            lines.append(indent + extra)
                for kk in early.get(k, ()):
                    if not emitted[kk]:
                        emit_node(kk)
                if k < len(self.nodes) and live[k] and not emitted[k]:
                    emit_node(k)
            return lines
        for i, (dst, r) in enumerate(self.outputs):
            if not isinstance(r, float):
                stack = [(abs(r), False)]
                while stack:
                    k, expanded = stack.pop()
                    if emitted[k]:
                        continue
                    if expanded:
                        emit_node(k)
                        continue
                    stack.append((k, True))
                    for d in reversed(deps(k)):
                        if not emitted[d]:
                            stack.append((d, False))
            lines.append(indent + store(dst, self._opnd(r, self._store_type(r))))
            if after_store is not None:
                extra = after_store(i)
                if extra:
                    lines.append(indent + extra)
        return lines

    def max_live(self):
        """Most values alive at once when the live part of the trace is emitted in creation order (the order of the column-serial and
        tile-cooperative cores): a lower bound of the registers the straight-line body needs -- hipcc adds addresses, temporaries of
        its own scheduling and, above the budget, spills.  Returns (count, node position of the maximum)."""
        live = self.live_nodes()
        last = {}
        for k in range(1, len(self.nodes)):
            if live[k]:
                for d in self._deps(k):
                    last[d] = k
        for (dst, r), pos in zip(self.outputs, self.out_pos):
            if not isinstance(r, float):
                last[abs(r)] = max(last.get(abs(r), 0), pos)
        ev = [0] * (len(self.nodes) + 2)
        for k in range(1, len(self.nodes)):
            if live[k] and self.nodes[k][0] != "lnd":
                ev[k] += 1
                ev[max(last.get(k, k), k) + 1] -= 1
        cur = mx = where = 0
        for i, e in enumerate(ev):
            cur += e
            if cur > mx:
                mx, where = cur, i
        return mx, where

    def _store_type(self, r):
        """An output is converted to the storage type T once, from whatever type its node has."""
        return 0 if isinstance(r, float) else self.ntype[abs(r)]

    def evaluate(self, inputs, dtype="float64"):
        """Interpret the live part of the trace with numpy (batched).  Used by the CPU-side tests to
        check a trace against the oracle and to study fp32 round-off without a compiler or a GPU.

        inputs: dict input-expression -> array (K,).  dtype float32 emulates fp32 storage with fused
        multiply-add (product and sum formed in float64, rounded once to float32).
        Returns the list of output arrays in ``self.outputs`` order.
        """
        import numpy as np
        dt = np.dtype(dtype)
        live = self.live_nodes()
        val = [None] * len(self.nodes)
        f32 = (dt != np.float64)

        def r32(x):
            return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)

        def get(r, hp=0):
            # operand as seen by an operation of type hp (0: C, 1: D); in float64 mode there is only one type
            if isinstance(r, float):
                return np.float64(r) if (hp or not f32) else np.float64(np.float32(r))
            x = val[abs(r)]
            if f32 and not hp and self.ntype[abs(r)]:
                x = r32(x)                       # narrowing conversion at the use
            return x if r > 0 else -x

        for k in range(1, len(self.nodes)):
            if not live[k]:
                continue
            op, a, b, c = self.nodes[k]
            hp = self.ntype[k]
            rnd = (lambda x: x) if (hp or not f32) else r32
            g = lambda r: get(r, hp)
            if op == "in" and (a.startswith("in.utab_get(") or a.startswith("in.m_get(")):
                # wave tables: the LATEST value written to the slot before this read (slots are rewritten); a matrix read takes
                # the value of the lane that owns the column (the batch axis of this evaluation is the lane axis)
                args = a[a.index("(") + 1:a.index(")")].split(",")
                dst = ("utab:" if a.startswith("in.utab_get(") else "mput:") + args[0]
                src = [r for (d, r), pos in zip(self.outputs, self.out_pos) if d == dst and pos <= k]
                if not src and dst.startswith("utab:") and "__utab_init__" in inputs:
                    val[k] = rnd(np.float64(inputs["__utab_init__"][int(args[0])]) + np.zeros(1))       # (written by another wave's trace)
                    continue
                assert src, "wave table slot %s read before it was written" % dst
                x = np.asarray(get(src[-1], 0 if isinstance(src[-1], float) else self.ntype[abs(src[-1])]), dtype=np.float64).reshape(-1)
                pick = x[int(args[1]) if (dst.startswith("mput:") and x.size > 1) else 0]
                val[k] = rnd(pick + np.zeros(1))
            elif op == "in" and (a.startswith("in.lane_tab(") or a.startswith("in.lane_tab2(")):
                # per-lane read of a wave's uniform table: word base + 6 * (the lane's joint) + r (inputs["__kcol__"]: joint per lane;
                # the helper role reads the OTHER wave's table: suffix 2, inputs["__kcol2__"]); slots this trace never wrote come from
                # inputs["__utab_init__"] (what another wave's trace wrote)
                sfx = "2" if a.startswith("in.lane_tab2(") else ""
                base, r = (int(x) for x in a[a.index("(") + 1:a.index(")")].split(","))
                kcol = np.asarray(inputs["__kcol%s__" % sfx], dtype=int)
                res = np.zeros(kcol.shape)
                for jj in np.unique(kcol):
                    slot = base + 6 * int(jj) + r
                    src = [rr for (d, rr), pos in zip(self.outputs, self.out_pos) if d == "utab%s:%d" % (sfx, slot) and pos <= k]
                    if src:
                        x = np.asarray(get(src[-1], 0 if isinstance(src[-1], float) else self.ntype[abs(src[-1])]), dtype=np.float64).reshape(-1)
                        res[kcol == jj] = x[0]
                    else:
                        res[kcol == jj] = inputs["__utab_init__"][slot]
                val[k] = rnd(res)
            elif op == "in" and a.startswith("in.tab_get("):
                slot = a[len("in.tab_get("):a.index(")")]
                src = [r for (dst, r) in self.outputs if dst == "tab:" + slot]
                assert len(src) == 1 and (isinstance(src[0], float) or abs(src[0]) < k), "table slot read before it was written"
                val[k] = rnd(get(src[0]) + np.zeros(1))
            elif op == "in" and a.startswith("in.xch_get("):
                # published by another wave's core: per slot, or per REQUEST (the full expression with its serial) where the caller
                # distinguishes what a slot held in the phase of this read from what it holds later (tests/test_lean.py)
                val[k] = rnd(np.asarray(inputs[a] if a in inputs else inputs[a.split("/*")[0]], dtype=np.float64))
            elif op == "in":
                val[k] = rnd(np.asarray(inputs[a], dtype=np.float64))
            elif op == "mul":
                val[k] = rnd(g(a) * g(b))
            elif op == "add":
                val[k] = rnd(g(a) + g(b))
            elif op == "fma":
                val[k] = rnd(g(a) * g(b) + g(c))
            elif op == "rcp":
                val[k] = rnd(1.0 / g(a))
            elif op == "pkfma":
                val[k] = (rnd(get(a[0]) * get(b[0]) + get(c[0])), rnd(get(a[1]) * get(b[1]) + get(c[1])))
            elif op == "pkmul":
                val[k] = (rnd(get(a[0]) * get(b[0])), rnd(get(a[1]) * get(b[1])))
            elif op == "pkadd":
                val[k] = (rnd(get(a[0]) + get(b[0])), rnd(get(a[1]) + get(b[1])))
            elif op == "lnd":
                val[k] = val[a]
            elif op == "bc":
                x = np.asarray(val[a], dtype=np.float64).reshape(-1)
                val[k] = x[b if x.size > 1 else 0] + np.zeros(1)
            elif op == "cvt":
                val[k] = rnd(val[a])
            elif op == "lo":
                val[k] = val[a][0]
            elif op == "hi":
                val[k] = val[a][1]
            elif op == "sin":
                val[k] = rnd(np.sin(g(a)))
            elif op == "cos":
                val[k] = rnd(np.cos(g(a)))
        outs = []
        for (_, r) in self.outputs:
            if isinstance(r, float):
                outs.append(np.float64(r))
            else:
                x = get(r, self.ntype[abs(r)])
                outs.append(r32(x) if f32 else x)     # stored as T
        return outs
