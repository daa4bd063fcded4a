"""The traced cores (what one lane executes) interpreted on the CPU against the oracle -- no compiler,
no GPU: float64 proves the algorithm + sparsity specialisation are exact; emulated fp32 (fused
multiply-add, float storage) bounds the round-off the HIP kernels will show."""
import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd.emit import cores
from gridcodegenerator_amd.emit.model import RobotSpec
from oracle import rbd_oracle as O

K = 6


def _inputs(n, q, qd, u=None, qdd=None, Minv_flat=None, style="core"):
    d = {"gravity": np.full(q.shape[0], 9.81)}
    for j in range(n):
        d["in.q(%d)" % j] = q[:, j]; d["in.qd(%d)" % j] = qd[:, j]
        if u is not None:
            d["in.u(%d)" % j] = u[:, j]
        if qdd is not None:
            d["in.qdd(%d)" % j] = qdd[:, j]
    if Minv_flat is not None:
        for i in range(n * n):
            d["in.Minv(%d)" % i] = Minv_flat[:, i]
    return d


def _run(tr, inputs, dtype="float64"):
    Kb = len(inputs["gravity"])
    return np.stack([np.broadcast_to(o, (Kb,)) for o in tr.evaluate(inputs, dtype)], axis=1)


def _grad_flat(M, n):
    return np.concatenate([O.flat_colmajor(M[:, :, :n]), O.flat_colmajor(M[:, :, n:])], axis=1)


@pytest.fixture(scope="module")
def setup(robots, tables):
    def make(name):
        spec = RobotSpec(robots(name))
        q, qd, u = (a.astype(np.float64) for a in make_inputs(spec.n, K, 11))
        return spec, tables(name), q, qd, u
    return make


def test_rnea_core(robot_name, setup):
    spec, T, q, qd, u = setup(robot_name)
    qdd = u * 0.7
    for use_qdd in (False, True):
        tr = cores.core_inverse_dynamics(spec, use_qdd)
        got = _run(tr, _inputs(spec.n, q, qd, qdd=qdd if use_qdd else None))
        ref = O.rnea(T, q, qd, qdd if use_qdd else None)[0]
        assert relerr(got, ref)[0] < 1e-13


def test_vaf_core(robot_name, setup):
    spec, T, q, qd, u = setup(robot_name)
    n = spec.n
    tr = cores.core_inverse_dynamics_vaf(spec, True)
    got = _run(tr, _inputs(n, q, qd, qdd=u))
    c, v, a, f = O.rnea(T, q, qd, u)
    ref = np.concatenate([v.reshape(K, 6 * n), a.reshape(K, 6 * n), f.reshape(K, 6 * n)], axis=1)
    assert relerr(got, ref)[0] < 1e-13


def test_minv_core(robot_name, setup):
    spec, T, q, qd, u = setup(robot_name)
    got = _run(cores.core_direct_minv(spec), _inputs(spec.n, q, qd))
    ref = O.flat_colmajor(O.minv(T, q, False))
    assert relerr(got, ref)[0] < 1e-13
    n = spec.n
    lower = [n * c + r for c in range(n) for r in range(n) if r > c]
    assert np.all(got[:, lower] == 0.0)  # upper triangular output, lower half written as 0


def test_forward_dynamics_core(robot_name, setup):
    spec, T, q, qd, u = setup(robot_name)
    got = _run(cores.core_forward_dynamics(spec), _inputs(spec.n, q, qd, u=u))
    assert relerr(got, O.forward_dynamics(T, q, qd, u))[0] < 1e-12


def test_rnea_gradient_core(robot_name, setup):
    spec, T, q, qd, u = setup(robot_name)
    for use_qdd in (False, True):
        tr = cores.core_inverse_dynamics_gradient(spec, use_qdd)
        got = _run(tr, _inputs(spec.n, q, qd, qdd=u if use_qdd else None))
        ref = _grad_flat(O.rnea_grad(T, q, qd, u if use_qdd else None), spec.n)
        assert relerr(got, ref)[0] < 1e-13


def test_fd_gradient_core(robot_name, setup):
    spec, T, q, qd, u = setup(robot_name)
    ref, parts = O.fd_grad(T, q, qd, u, return_parts=True)
    ref = _grad_flat(ref, spec.n)
    got = _run(cores.core_forward_dynamics_gradient(spec, False), _inputs(spec.n, q, qd, u=u))
    assert relerr(got, ref)[0] < 1e-12
    # variant with qdd and upper-triangular Minv supplied (reference USE_QDD_MINV_FLAG)
    Mup = O.flat_colmajor(np.triu(parts["Minv"]))
    got2 = _run(cores.core_forward_dynamics_gradient(spec, True), _inputs(spec.n, q, qd, qdd=parts["qdd"], Minv_flat=Mup))
    assert relerr(got2, ref)[0] < 1e-12


def test_structural_zeros_are_folded(robots):
    """dc_du[j, col] vanishes unless col is an ancestor of j or in its subtree; the generator must know it."""
    spec = RobotSpec(robots("atlas30"))
    tr = cores.core_inverse_dynamics_gradient(spec, True)
    n = spec.n
    const_zero = sum(1 for (_, r) in tr.outputs if isinstance(r, float) and r == 0.0)
    related = sum(len(spec.ancestors[j]) + len(spec.subtree[j]) for j in range(n))
    assert const_zero >= 2 * (n * n - related)  # (+ columns the dynamics is invariant to, e.g. base yaw)


def test_op_counts_beat_dense(robots):
    """Generation-time sparsity: far fewer flops than the dense 6x6 count of SURVEY.md section 8(d)."""
    assert cores.core_forward_dynamics_gradient(RobotSpec(robots("iiwa7")), False).flops() < 0.5 * 41000
    assert cores.core_forward_dynamics_gradient(RobotSpec(robots("atlas30")), False).flops() < 0.5 * 337000


@pytest.mark.parametrize("name,bound", [("iiwa7", 3e-6), ("atlas30", 1e-5)])
def test_fp32_roundoff_budget(name, bound, setup):
    """Emulated fp32: the norm-wise error of df_du stays at the level of the reference's own fp32
    kernels or better (2.4e-5 / 5.9e-5)."""
    spec, T, q, qd, u = setup(name)
    ref = _grad_flat(O.fd_grad(T, q, qd, u), spec.n)
    got = _run(cores.core_forward_dynamics_gradient(spec, False), _inputs(spec.n, q, qd, u=u), "float32")
    assert relerr(got, ref)[0] < bound


# ---- explicit schedules for large robots: two-pass (workspace) cores and the recomputing single-kernel cores ----------
def _scatter(tr, outs, width):
    got = np.zeros((outs.shape[0], width))
    for (dst, _), col in zip(tr.outputs, outs.T):
        got[:, int(dst)] = col
    return got


@pytest.mark.parametrize("kind", ["id", "id_qdd", "fd"])
def test_two_pass_cores(robot_name, setup, kind):
    """prep core -> workspace values -> column-serial core == oracle (what *_prep_kernel + *_columns_kernel compute)."""
    import re
    spec, T, q, qd, u = setup(robot_name)
    n = spec.n
    fd = (kind == "fd")
    ws = cores.WorkspaceMap(spec, with_minv=fd)
    prep = cores.core_gradient_prep(spec, ws, "fd" if fd else "id", use_qdd=(kind == "id_qdd"))
    wsv = _scatter(prep, _run(prep, _inputs(n, q, qd, u=u, qdd=u)), ws.count)
    cols = cores.core_gradient_columns(spec, ws, fd)
    inputs = _inputs(n, q, qd)
    for k in range(1, len(cols.nodes)):
        op, a = cols.nodes[k][0], cols.nodes[k][1]
        if op == "in" and a.startswith("in.ws("):
            inputs[a] = wsv[:, int(re.match(r"in\.ws\((\d+)\)", a).group(1))]
    got = _scatter(cols, _run(cols, inputs), 2 * n * n)
    if fd:
        ref = _grad_flat(O.fd_grad(T, q, qd, u), n)
    else:
        ref = _grad_flat(O.rnea_grad(T, q, qd, u if kind == "id_qdd" else None), n)
    assert relerr(got, ref)[0] < 1e-12


@pytest.mark.parametrize("variant", ["id", "id_qdd", "fd", "fd_qdd_minv"])
def test_recompute_cores(robot_name, setup, variant):
    """Column-serial cores that recompute v, a, f per column (single-kernel schedule for n > 12) == oracle."""
    spec, T, q, qd, u = setup(robot_name)
    n = spec.n
    ref_fd, parts = O.fd_grad(T, q, qd, u, return_parts=True)
    if variant.startswith("id"):
        use_qdd = variant.endswith("qdd")
        tr = cores.core_gradient_recompute(spec, "id", use_qdd=use_qdd)
        inputs = _inputs(n, q, qd, qdd=u if use_qdd else None)
        ref = _grad_flat(O.rnea_grad(T, q, qd, u if use_qdd else None), n)
    elif variant == "fd":
        tr = cores.core_gradient_recompute(spec, "fd")
        inputs = _inputs(n, q, qd, u=u)
        ref = _grad_flat(ref_fd, n)
    else:
        tr = cores.core_gradient_recompute(spec, "fd", use_qdd_minv=True)
        inputs = _inputs(n, q, qd, qdd=parts["qdd"], Minv_flat=O.flat_colmajor(np.triu(parts["Minv"])))
        ref = _grad_flat(ref_fd, n)
    got = _scatter(tr, _run(tr, inputs), 2 * n * n)
    assert relerr(got, ref)[0] < 1e-12
    assert tr.op_counts().get("lnd", 0) > 0      # inputs are laundered per column (keeps hipcc from undoing the recomputation)


@pytest.mark.parametrize("variant", ["id", "fd", "fd_qdd_minv"])
def test_recompute_cores_with_table_and_column_groups(robot_name, setup, variant):
    """Options of the recomputing schedule: inputs parked in the lane-private table (tab_put / tab_get, resolved by the IR
    interpreter) and column groups written at local indices (the column-split kernels of large robots)."""
    spec, T, q, qd, u = setup(robot_name)
    n = spec.n
    ref_fd, parts = O.fd_grad(T, q, qd, u, return_parts=True)
    if variant == "id":
        kw, inputs, ref = dict(kind="id"), _inputs(n, q, qd), O.rnea_grad(T, q, qd, None)
    elif variant == "fd":
        kw, inputs, ref = dict(kind="fd"), _inputs(n, q, qd, u=u), ref_fd
    else:
        kw = dict(kind="fd", use_qdd_minv=True)
        inputs, ref = _inputs(n, q, qd, qdd=parts["qdd"], Minv_flat=O.flat_colmajor(np.triu(parts["Minv"]))), ref_fd
    tr = cores.core_gradient_recompute(spec, table=True, **kw)
    outs = _run(tr, inputs)
    got = np.zeros((K, 2 * n * n))
    ntab = 0
    for (dst, _), col in zip(tr.outputs, outs.T):
        if isinstance(dst, str):
            ntab += 1
        else:
            got[:, dst] = col
    assert ntab > 0 and ntab <= cores.recompute_table_size(spec, kw["kind"], use_qdd=True)
    assert relerr(got, _grad_flat(ref, n))[0] < 1e-12
    # a column group: d/dq columns first, then d/dqd, at local indices
    cols = list(range(n // 3, 2 * n // 3 + 1))
    trc = cores.core_gradient_recompute(spec, cols=cols, **kw)
    gotc = _scatter(trc, _run(trc, inputs), 2 * n * len(cols))
    want = np.concatenate([ref[:, :, c] for c in cols] + [ref[:, :, n + c] for c in cols], axis=1)
    assert relerr(gotc, want)[0] < 1e-12


def test_packed_pair_traces(setup):
    """EXPERIMENTAL packed=True emission (the d/dq and d/dqd recursions as v_pk_fma_f32 pairs): same values, fewer
    instructions.  Not shipped (hipcc allocates the 64-bit pairs badly) but kept correct."""
    from gridcodegenerator_amd.emit.trace import Tracer
    spec, T, q, qd, u = setup("iiwa7")
    plain = cores.core_forward_dynamics_gradient(spec, False)
    assert not any(k.startswith("pk") for k in plain.op_counts())
    Tracer.use_packed = True
    try:
        packed = cores.core_forward_dynamics_gradient(spec, False)
    finally:
        Tracer.use_packed = False
    assert packed.op_counts().get("pkfma", 0) > 1000 and packed.arith_instructions() < 0.8 * plain.arith_instructions()
    ref = _grad_flat(O.fd_grad(T, q, qd, u), spec.n)
    assert relerr(_run(packed, _inputs(spec.n, q, qd, u=u)), ref)[0] < 1e-12
    assert relerr(_run(plain, _inputs(spec.n, q, qd, u=u)), ref)[0] < 1e-12


def test_optimal_column_sets(setup):
    """Exhaustive column-set partition: exact costs (equal to re-tracing the group), never worse than the best contiguous
    split, every column exactly once."""
    spec, T, q, qd, u = setup("iiwa7")
    builder = lambda cols: cores.core_forward_dynamics_gradient(spec, False, cols)
    full = builder(None)
    contiguous_cost = cores.range_cost_function(spec, builder, exact=True)
    for S in (3, 4):
        parts, worst = cores.optimal_column_sets(spec, S, full)
        assert sorted(c for p in parts for c in p) == list(range(spec.n)) and len(parts) == S
        assert worst == max(cores._arith_ops(builder(p)) for p in parts)
        assert worst <= cores.balanced_column_split(spec, S, contiguous_cost)[1]
    # a non-contiguous group evaluates to the same numbers as the full gradient restricted to its columns
    cols = [0, 4]
    got = _scatter(builder(cols), _run(builder(cols), _inputs(spec.n, q, qd, u=u)), 2 * spec.n * len(cols))
    ref = O.fd_grad(T, q, qd, u)
    want = np.concatenate([ref[:, :, c] for c in cols] + [ref[:, :, spec.n + c] for c in cols], axis=1)
    assert relerr(got, want)[0] < 1e-12


def test_reference_prismatic_gradient_option(setup, golden, monkeypatch):
    """GRiDCodeGenerator(prismatic_gradient="reference"): the d/dq seed of a joint's own column uses the reference's motion cross
    product (_test.py:311,437).  The traced gradient cores of the prismatic test robot then reproduce what the REFERENCE ITSELF
    computed (tests/golden/mixed5.npz: dc_du, df_du) -- which the default ("corrected": force cross product, the one finite differences
    confirm) deliberately does not -- and for a revolute-only robot the two options trace the very same operations."""
    from gridcodegenerator_amd.emit import algorithms as alg
    spec, T, q, qd, u = setup("mixed5")
    n = spec.n
    Gd = golden("mixed5")
    qg, qdg, ug = Gd["q"], Gd["qd"], Gd["u"]
    monkeypatch.setattr(alg, "PRISMATIC_GRADIENT", "reference")
    got = _run(cores.core_inverse_dynamics_gradient(spec, False), _inputs(n, qg, qdg))
    assert relerr(got, _grad_flat(Gd["dc_du_noqdd"], n))[0] < 1e-11
    assert relerr(got, _grad_flat(O.rnea_grad(T, qg, qdg, None, prismatic_fix=False), n))[0] < 1e-12
    got_fd = _run(cores.core_forward_dynamics_gradient(spec, False), _inputs(n, qg, qdg, u=ug))
    assert relerr(got_fd, _grad_flat(Gd["df_du"], n))[0] < 1e-9
    # the column-serial schedule (what the mixed5 library ships) takes the same switch
    tr_cs = cores.core_gradient_recompute(spec, "id")
    vals = tr_cs.evaluate(_inputs(n, qg, qdg))
    got_cs = np.zeros((qg.shape[0], 2 * n * n))
    for (dst, _), v in zip(tr_cs.outputs, vals):            # (column-serial cores emit the columns in their own order)
        if not isinstance(dst, str):
            got_cs[:, int(dst)] = v
    assert relerr(got_cs, _grad_flat(Gd["dc_du_noqdd"], n))[0] < 1e-11
    monkeypatch.setattr(alg, "PRISMATIC_GRADIENT", "corrected")
    corrected = _run(cores.core_inverse_dynamics_gradient(spec, False), _inputs(n, qg, qdg))
    assert relerr(corrected, _grad_flat(Gd["dc_du_noqdd"], n))[0] > 1e-3            # (the reference's prismatic columns differ)
    assert relerr(corrected, _grad_flat(O.rnea_grad(T, qg, qdg, None, prismatic_fix=True), n))[0] < 1e-12
    # revolute-only robot: the same trace either way
    spec7 = RobotSpec(__import__("gridcodegenerator_amd.robots", fromlist=["get_robot"]).get_robot("iiwa7"))
    a = cores.core_inverse_dynamics_gradient(spec7, False)
    monkeypatch.setattr(alg, "PRISMATIC_GRADIENT", "reference")
    b = cores.core_inverse_dynamics_gradient(spec7, False)
    assert a.nodes == b.nodes and a.outputs == b.outputs
