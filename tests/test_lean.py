"""Register-lean tile-cooperative forward-dynamics-gradient kernel (`forward_dynamics_gradient_kernel_coop8`): EIGHT wavefronts per
tile of 64 configurations -- two per SIMD, at most 256 registers each -- instead of four that own a SIMD and 464 registers.  What
makes the cores lean (emit/cores.py: CoopSlots.enable_lean, LeanRole, lean_plan; emit/algorithms.py: direct_minv_lean): a
block-shared input table, the Minv recursion by base-rooted tree with U, 1/D and the backward-pass entries parked in LDS, bias
torques depth first, qdd rows after the barrier, X rebuilt for the way back up the tree, and gradient HALF columns as the unit of
work.  The block's cores are interpreted here on the CPU (numpy over the tracer IR) with the exchange region as a dictionary.
Reference mapping being replaced: algorithms/_forward_dynamics_gradient.py:7-57, algorithms/_inverse_dynamics_gradient.py:199-246,501-540."""
import os

import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd.emit import cores
from gridcodegenerator_amd.emit.model import RobotSpec


def emulate_lean_block(spec, slots, plan, q, qd, u, kind="fd", qdd=None, dtype="float64"):
    """The block's barriers separate its phases; phase p of every wave reads only what phases < p published.  Sweep p evaluates every
    wave's core against the exchange region as it stands and publishes the `xch` writes of PHASE p only (a Minv slot is written
    twice -- the backward pass's value before B1, the final value between B1 and B2 -- and the forward pass must read the former);
    the outputs are taken from the last sweep."""
    n, K = spec.n, q.shape[0]
    traces = [cores.core_gradient_recompute(spec, kind, use_qdd=qdd is not None, cols=items, coop=(role, slots)) for (role, items) in plan]
    base = {"gravity": np.full(K, 9.81)}
    for j in range(n):
        base["in.q(%d)" % j] = q[:, j]; base["in.qd(%d)" % j] = qd[:, j]; base["in.u(%d)" % j] = u[:, j]
        if qdd is not None:
            base["in.qdd(%d)" % j] = qdd[:, j]
    xch = {"in.xch_get(%d)" % s: np.zeros(K) for s in range(-8 * 34, slots.count)}     # (negative: the parking words below the region)
    got = np.full((K, 2 * n * n), np.nan)
    phases = [d for (d, _) in traces[0].outputs].count("barrier") + 1       # (4 barriers; 5 in the mixed arithmetic; 1 for the dID block)
    assert all([d for (d, _) in tr.outputs].count("barrier") == phases - 1 for tr in traces)
    # every exchange read of a trace with the phase it sits in: in sweep p a read of an EARLIER phase must see what its slot held then
    # (a Minv slot holds the backward-pass value before B1 and the final value after B2; values computed from the former and
    # published later -- the double partial sums of qdd in the mixed arithmetic -- would otherwise be recomputed from the latter)
    reads = []
    for tr in traces:
        bars = sorted(pos for (d, _), pos in zip(tr.outputs, tr.out_pos) if d == "barrier")
        live = tr.live_nodes()
        reads.append([(tr.nodes[k][1], sum(1 for b in bars if b <= k)) for k in range(1, len(tr.nodes))
                      if live[k] and tr.nodes[k][0] == "in" and isinstance(tr.nodes[k][1], str) and tr.nodes[k][1].startswith("in.xch_get(")])
    states = []
    with np.errstate(all="ignore"):
        for sweep in range(phases):
            new = {}
            states.append(dict(xch))
            for tr, rd in zip(traces, reads):
                inp = dict(base); inp.update(xch)
                for (expr, ph) in rd:
                    if ph < sweep:
                        inp[expr] = states[ph][expr.split("/*")[0]]
                outs = tr.evaluate(inp, dtype=dtype)
                phase = 0
                piece = {}
                for (dst, _), o in zip(tr.outputs, outs):
                    if dst == "barrier":
                        phase += 1
                    elif isinstance(dst, str) and dst.startswith("xch:"):
                        if phase == sweep:
                            new["in.xch_get(%s)" % dst[4:]] = np.broadcast_to(o, (K,)).astype(np.float32).astype(np.float64)
                    elif isinstance(dst, str) and dst.startswith("piece:"):          # (AlignedPieces: values of a piece, then its flush)
                        piece[int(dst.split(":")[2])] = o
                    elif isinstance(dst, str) and dst.startswith("flush:"):
                        _, length, offset = dst.split(":")
                        assert sorted(piece) == list(range(int(length)))
                        if sweep == phases - 1:
                            for pos, val in piece.items():
                                assert np.isnan(got[:, int(offset) + pos]).all(), "an output written twice"
                                got[:, int(offset) + pos] = val
                        piece = {}
                    elif not isinstance(dst, str) and sweep == phases - 1:
                        run, r = divmod(int(dst), n)
                        got[:, tr.run_bases[run] + r] = o
                assert not piece
            xch.update(new)
    emulate_lean_block.last_exchange = xch          # (for debugging: the block's LDS after the last phase)
    return got, traces


SCATTERED = dict(order="lpt", umc=False, aligned_flush=False, chain_f=False)
# options of the planner / the cores that were measured on the GPU and NOT shipped (profiles/r04/): their emulation tests run on request
# (GRID_TEST_EXPERIMENTS=1) so that the default CPU suite stays short
experiment = pytest.mark.skipif(os.environ.get("GRID_TEST_EXPERIMENTS", "0") != "1", reason="rejected experiment: set GRID_TEST_EXPERIMENTS=1")       # the first form of the planner / the sink (kept as options)


@pytest.mark.parametrize("robot,options", [("atlas30", {}), ("atlas30", SCATTERED), ("mixed5", {}), ("iiwa7", {})],
                         ids=["atlas30", "atlas30-scattered", "mixed5", "iiwa7"])
def test_lean_block_matches_oracle(robot, options, robots, tables):
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(robot))
    n, K = spec.n, 3
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 61))
    ref = O.fd_grad(tables(robot), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec, **options)
    got, traces = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got).any()                                      # every one of the 2 n^2 outputs is written by some wave
    assert relerr(got, ref)[0] < 5e-6                                   # (Minv, c, qdd cross the exchange region as fp32)
    # every half-column exactly once over the block; every wave executes the same three barriers
    items = sorted(it for (_, its) in plan for it in its)
    assert items == [(c, h) for c in range(n) for h in (0, 1)]
    for tr in traces:
        assert [d for (d, _) in tr.outputs].count("barrier") == cores.LEAN_BARRIERS
    # the exchange region is written consistently: a slot has ONE publishing wave per phase (a Minv slot: the backward pass's wave
    # before B1, the wave that owns its column between B1 and B2)
    writers = {}
    for w, tr in enumerate(traces):
        phase = 0
        for (d, _) in tr.outputs:
            phase += (d == "barrier")
            if isinstance(d, str) and d.startswith("xch:"):
                writers.setdefault((int(d[4:]), phase), set()).add(w)
    # (the parking words below the region -- U, 1/D of the articulated-inertia chain -- may have two publishers: the two waves that
    #  divide the largest tree's backward pass both run the chain and write the same values)
    assert all(len(ws) == 1 for (s_, ph_), ws in writers.items() if s_ >= 0), "a slot with two publishers in one phase"
    writers = {s_: ws for (s_, ph_), ws in writers.items()}
    table = set(slots.itab["qd"]) | set(slots.itab.get("u", []))
    for j in range(n):
        table |= {slots.itab["s"][j], slots.itab["c"][j]} if spec.uses_trig[j] else {slots.itab["q"][j]}
    assert table <= set(writers) and set(slots.c) <= set(writers) and set(slots.qdd) <= set(writers) and set(slots.minv.values()) <= set(writers)


@experiment
def test_lean_block_with_per_column_minv_matches_oracle(robots, tables):
    """lean_plan(columns_from_chain=True): only the articulated-inertia chain (U, 1/D) of the Minv recursion is serial and published;
    the backward pass's F recursions move into the per-column phase (alg.minv_columns_lean) that all eight waves share."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 62))
    ref = O.fd_grad(tables("atlas30"), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec, columns_from_chain=True, **SCATTERED)
    got, traces = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got).any() and relerr(got, ref)[0] < 5e-6
    assert sorted(k for (role, _) in plan for k in role.minv_cols) == list(range(n))


@pytest.mark.parametrize("robot", ["atlas30", "quad12"])
def test_lean_block_in_contiguous_runs_matches_oracle(robot, robots, tables):
    """lean_plan(order="runs"): every wave takes one contiguous run of d/dq columns and one of d/dqd columns (the half-columns it
    flushes one after the other are neighbours in the configuration's output row); same outputs, every half-column exactly once,
    parked columns inside the wave's own d/dqd run."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(robot))
    n, K = spec.n, 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 63))
    ref = O.fd_grad(tables(robot), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec, order="runs", umc=False, aligned_flush=False, chain_f=False)
    got, traces = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got).any() and relerr(got, ref)[0] < 5e-6
    seen = []
    for role, items in plan:
        for h in (0, 1):
            cols = sorted(c for (c, h_) in items if h_ == h)
            assert cols == list(range(cols[0], cols[-1] + 1)) if cols else True       # one contiguous run per block
            seen += [(c, h) for c in cols]
        assert set(role.hoist) <= set(c for (c, h_) in items if h_ == 1)
    assert sorted(seen) == [(c, h) for c in range(n) for h in (0, 1)]
    lpt = cores.lean_plan(spec, **SCATTERED)[0]
    assert max(slots.lean_model["post"]) <= 1.05 * max(lpt.lean_model["post"])       # balanced nearly as well as the scattered sets


def test_lean_block_with_sector_aligned_pieces(robots, tables):
    """lean_plan(order="runs", aligned_flush=True, umc=True): the output leaves in pieces cut at the 32-byte sectors of the row (at
    most 32 values, the remainder carried into the next half-column's piece); u - c instead of c and u in LDS."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 64))
    ref = O.fd_grad(tables("atlas30"), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec, order="runs", aligned_flush=True, umc=True, chain_f=False)
    assert "u" not in slots.itab and min(slots.c) == -8 * n
    got, traces = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got).any() and relerr(got, ref)[0] < 5e-6
    partial = total = 0
    for tr in traces:
        for (d, _) in tr.outputs:
            if isinstance(d, str) and d.startswith("flush:"):
                _, length, base = (int(x) if x.isdigit() else x for x in d.split(":"))
                assert 1 <= length <= 32
                total += 1
                partial += (base % 8 != 0) + ((base + length) % 8 != 0)
    # a half-column flushed as it is writes into two partial sectors (2 x 60 here); cut at the sectors only the ends of the 16 runs do
    assert total <= 2 * n + 16 and partial <= 2 * 16, (total, partial)
    for (role, items) in plan:              # the parked columns are the first d/dqd columns of the wave's run
        hi = sorted(c for (c, h) in items if h == 1)
        assert role.hoist == hi[:len(role.hoist)]


@pytest.mark.parametrize("use_qdd", [False, True])
def test_lean_inverse_dynamics_gradient_block_matches_oracle(use_qdd, robots, tables):
    """The register-lean block of the INVERSE-dynamics gradient (cores.lean_plan_id: input table, one barrier, contiguous runs of
    gradient half-columns, sector-aligned pieces) against the oracle's rnea_grad, with and without a given qdd."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 65))
    qdd = np.random.default_rng(66).uniform(-1.0, 1.0, (K, n)) if use_qdd else None
    dc = O.rnea_grad(tables("atlas30"), q, qd, qdd if use_qdd else np.zeros((K, n)))
    ref = np.concatenate([O.flat_colmajor(dc[:, :, :n]), O.flat_colmajor(dc[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan_id(spec, use_qdd)
    got, traces = emulate_lean_block(spec, slots, plan, q, qd, u, kind="id", qdd=qdd)
    assert not np.isnan(got).any() and relerr(got, ref)[0] < 2e-6           # (sin, cos, qd cross the input table as fp32)
    assert sorted(it for (_, its) in plan for it in its) == [(c, h) for c in range(n) for h in (0, 1)]
    for tr in traces:
        assert [d for (d, _) in tr.outputs].count("barrier") == 1
        assert tr.max_live()[0] <= 128
    assert 4 * 64 * (slots.count + cores.LEAN_WAVES * 34) <= 160 * 1024     # (100 KB: one block of 8 x 256 registers fills a CU anyway)


def test_lean_runs_walked_towards_the_root_reuse_the_childs_force(robots):
    """chain_f (shipped in the forward-dynamics-gradient kernel): the d/dq run of a wave is walked from its deepest column towards
    the root, and a column whose child was finished just before takes the child's accumulated force from registers instead of walking
    the child's subtree again -- fewer instructions, same outputs (test_lean_block_matches_oracle runs with it), no more live values."""
    spec = RobotSpec(robots("atlas30"))
    work = {}
    for chain in (False, True):
        slots, plan = cores.lean_plan(spec, chain_f=chain)
        traces = [cores.core_gradient_recompute(spec, "fd", cols=items, coop=(role, slots)) for (role, items) in plan]
        work[chain] = (sum(cores.lean_arith(tr) for tr in traces), max(tr.max_live()[0] for tr in traces))
        if chain:
            for tr in traces:           # the d/dq block (row offsets below n*n) leaves from the top of the wave's run downwards
                lo = [int(d.split(":")[2]) for (d, _) in tr.outputs if isinstance(d, str) and d.startswith("flush:") and int(d.split(":")[2]) < spec.n ** 2]
                assert not lo or (lo[0] >= max(lo) - 32 and lo[-1] <= min(lo) + 32)
    assert work[True][0] < 0.95 * work[False][0] and work[True][1] <= work[False][1] + 8, work


@experiment
def test_lean_block_with_paired_products_matches_oracle(robots, tables):
    """lean_plan(pair_products=True) -- measured slower on the GPU and not shipped (profiles/r04/exp_pair_products.txt), kept as an
    option: two half-columns of one tree share one -Minv dc product with packed multiply-adds; same outputs, fewer instructions."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 67))
    ref = O.fd_grad(tables("atlas30"), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec, pair_products=True)
    got, traces = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got).any() and relerr(got, ref)[0] < 5e-6
    packed = sum(1 for tr in traces for k, live in enumerate(tr.live_nodes()) if live and k and tr.nodes[k][0] == "pkfma")
    assert packed > 2000


@experiment
def test_lean_block_in_the_mixed_arithmetic(robots, tables, monkeypatch):
    """Experimental `lean_mixed` (not in the shipped mixed library: profiles/r04/mixed_lean_report.txt): the Minv passes in double
    inside the waves, floats across LDS, qdd summed in double from per-wave partial sums published as float pairs (one more
    barrier).  Emulated in float32 storage: clearly better than the fp32 block on the same inputs, five barriers, the partial sums
    and the base joints' low parts fit the words below the exchange region."""
    from gridcodegenerator_amd.emit.trace import Tracer
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 64
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 5))
    ref = O.fd_grad(tables("atlas30"), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    err = {}
    for mixed in (False, True):
        monkeypatch.setattr(Tracer, "mixed", mixed)
        slots, plan = cores.lean_plan(spec)
        got, traces = emulate_lean_block(spec, slots, plan, q, qd, u, dtype="float32")
        assert not np.isnan(got).any()
        err[mixed] = relerr(got, ref)[0]
        assert [d for (d, _) in traces[0].outputs].count("barrier") == (5 if mixed else cores.LEAN_BARRIERS)
    assert slots.scratch_words <= cores.LEAN_WAVES * 34 and 2 * slots.partial_pairs <= 8 * n
    assert err[True] < 0.5 * err[False] and err[True] < 2.5e-6, err


def test_lean_forward_dynamics_block_matches_oracle(robots, tables):
    """cores.lean_plan_fd: the register-lean block as a forward-dynamics kernel -- the gradient kernel's prefix, no gradient columns, no
    parked recursions, wave 0 writes the n accelerations in one piece after B3."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 3
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 68))
    parts = O.fd_grad(tables("atlas30"), q, qd, u, return_parts=True)[1]
    slots, plan = cores.lean_plan_fd(spec)
    assert [role.out_qdd for (role, _) in plan] == [True] + [False] * 7 and all(not items and not role.hoist for (role, items) in plan)
    traces = [cores.core_gradient_recompute(spec, "fd", cols=items, coop=(role, slots)) for (role, items) in plan]
    got, _ = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got[:, :n]).any() and np.isnan(got[:, n:]).all()         # exactly the n accelerations are written (row offsets 0 .. n-1)
    assert relerr(got[:, :n], parts["qdd"])[0] < 1e-6
    total = sum(cores.lean_arith(tr) for tr in traces)
    assert total < 20000, total                # (a quarter of the gradient kernel's tile: 15.7 k instructions for Atlas-30)


def test_lean_direct_minv_block_matches_oracle(robots, tables):
    """cores.lean_plan_minv: the register-lean block as a direct-Minv kernel -- table of sin / cos only (the cores read nothing but q),
    no bias torques, no qdd; every wave writes the columns of its forward pass, upper triangle, the rest of the n x n output zero."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 3
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 69))
    Minv = O.fd_grad(tables("atlas30"), q, qd, u, return_parts=True)[1]["Minv"]
    ref = O.flat_colmajor(np.triu(Minv))
    slots, plan = cores.lean_plan_minv(spec)
    traces = [cores.core_gradient_recompute(spec, "fd", cols=items, coop=(role, slots)) for (role, items) in plan]
    for tr in traces:           # nothing but q is read from the input rows (stride_q may be NUM_JOINTS)
        live = tr.live_nodes()
        reads = set(tr.nodes[k][1].split("(")[0] for k in range(1, len(tr.nodes)) if live[k] and tr.nodes[k][0] == "in" and not tr.nodes[k][1].startswith("in.xch"))
        assert reads <= {"in.q"}, reads
    got, _ = emulate_lean_block(spec, slots, plan, q, qd, u)
    assert not np.isnan(got[:, :n * n]).any() and np.isnan(got[:, n * n:]).all()
    assert relerr(got[:, :n * n], ref)[0] < 1e-6
    assert np.all(got[:, :n * n][:, np.abs(ref).max(axis=0) == 0.0] == 0.0)


def test_lean_cores_stay_within_half_a_simd(robots):
    """The point of the exercise: the values a lean core holds at once (creation-order emission, the order the kernel is emitted in)
    stay far below 256 -- the 4-wave cores of the same robot hold 280-390 in their prologue alone -- and the block's LDS fits the CU."""
    spec = RobotSpec(robots("atlas30"))
    slots, plan = cores.lean_plan(spec)

    max_live = lambda tr: tr.max_live()[0]
    worst = max(max_live(cores.core_gradient_recompute(spec, "fd", cols=items, coop=(role, slots))) for (role, items) in plan)
    assert worst <= 160, worst
    old = cores.CoopSlots(spec); old.ksplit = 15
    assert max_live(cores.core_gradient_recompute(spec, "fd", cols=[29], coop=("producer2", old))) > 256
    assert 4 * 64 * (slots.count + cores.LEAN_WAVES * 34) <= 160 * 1024      # (staging rows of 34 words: pieces of 32 values)
    assert 8 * spec.n <= cores.LEAN_WAVES * 34                          # U, 1/D and u - c fit the staging regions below the exchange region


@pytest.mark.gpu
def test_lean_kernel_on_gpu(tables):
    """Through the C ABI (grid_set_coop mode 3): the 8-wave register-lean kernel against the oracle and against the 4-wave tile-cooperative
    kernel, ragged batches, few blocks (grid-stride over tiles: the exchange region and the parked U / 1/D are rewritten per tile), rows
    past the batch untouched; 0 B of scratch at <= 256 registers (two waves per SIMD is the point)."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    robot = "atlas30"
    host.build_library(robot, host.DEFAULT_PRECISION)
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        assert h.lean_available(host.ALG_FD_DU)
        attrs = h.L.kernel_attributes(host.ALG_FD_DU, coop=2)
        assert attrs["numRegs"] <= 256 and attrs["maxThreadsPerBlock"] >= 512 and attrs["scratch_bytes_per_lane"] == 0, attrs
        n = h.n
        for K in (1, 70, 333, 1500):
            q, qd, u = make_inputs(n, K, 90 + K)
            ref = oracle_all(T, q, qd, u)
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            h.set_coop(host.ALG_FD_DU, 2); h.set_wave(host.ALG_FD_DU, 1)
            four = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.forward_dynamics_gradient_device(four.data_ptr(), d_in.data_ptr(), 3 * n, K)
            h.synchronize()
            h.set_coop(host.ALG_FD_DU, 3)
            assert h.get_coop(host.ALG_FD_DU, K) == 2
            outs = []
            for blocks in (0, 1, 2):
                out = torch.full((K + 2, 2 * n * n), 4.25, dtype=torch.float32, device="cuda")
                h.forward_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks)
                h.synchronize()
                o = out.cpu().numpy()
                assert np.all(o[K:] == 4.25)
                outs.append(o[:K])
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
            err = relerr(outs[0], ref["df_du"])[0]
            print("lean kernel K=%d: df_du error %.2e (4-wave kernel %.2e)" % (K, err, relerr(four.cpu().numpy(), ref["df_du"])[0]))
            assert err < TOL[robot]["df_du"] * (4 if K < 64 else 1), (K, err)
            assert relerr(outs[0], four.cpu().numpy().astype(np.float64))[0] < 2 * TOL[robot]["df_du"]
            zero = np.abs(ref["df_du"]).max(axis=0) == 0.0
            assert np.all(outs[0][:, zero] == 0.0)
        h.set_coop(host.ALG_FD_DU, 0); h.set_wave(host.ALG_FD_DU, 0)


@pytest.mark.gpu
def test_lean_kernel_full_size_atlas30_16384(tables):
    """north_star's Atlas-30 batch of 16384 through the 8-wave kernel: spread sample against the oracle + permutation invariance."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    with host.GridHandle("atlas30", device=0, precision=host.DEFAULT_PRECISION) as h:
        n, K = h.n, 16384
        h.set_coop(host.ALG_FD_DU, 3)
        q, qd, u = make_inputs(n, K, 3)
        x = pack(q, qd, u)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3 * n, K)
        h.synchronize()
        df = d_out.cpu().numpy()
        rows = np.unique(np.concatenate([np.arange(64), np.linspace(0, K - 1, 96).astype(int), np.arange(K - 64, K)]))
        ref = oracle_all(tables("atlas30"), q[rows], qd[rows], u[rows])
        assert relerr(df[rows], ref["df_du"])[0] < TOL["atlas30"]["df_du"]
        perm = np.random.default_rng(4).permutation(K)
        d_in_p = torch.from_numpy(np.ascontiguousarray(x[perm])).cuda()
        d_out_p = torch.empty_like(d_out)
        h.forward_dynamics_gradient_device(d_out_p.data_ptr(), d_in_p.data_ptr(), 3 * n, K)
        h.synchronize()
        assert np.array_equal(d_out_p.cpu().numpy(), df[perm])


@pytest.mark.gpu
def test_lean_inverse_dynamics_gradient_kernel_on_gpu(tables):
    """`inverse_dynamics_gradient_kernel_coop8` through the C ABI: automatic for Atlas-30 at qdd = 0 (grid_get_coop = 2), against the
    oracle and against the lane-per-configuration kernel (other instruction order: to the parity tolerance); ragged batches, few blocks (grid-stride over tiles), rows past the batch untouched, a strided input
    (2n values used of rows of 3n), 0 B of scratch at <= 256 registers; a call with a given qdd keeps the other kernel."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    robot, alg = "atlas30", host.ALG_ID_DU
    host.build_library(robot, host.DEFAULT_PRECISION)
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        assert h.lean_available(alg)
        attrs = h.L.kernel_attributes(alg, coop=2)
        assert attrs["numRegs"] <= 256 and attrs["maxThreadsPerBlock"] >= 512 and attrs["scratch_bytes_per_lane"] == 0, attrs
        n = h.n
        for K in (1, 70, 333, 1500):
            q, qd, u = make_inputs(n, K, 190 + K)
            ref = oracle_all(T, q, qd, u)["dc_du_noqdd"]
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            h.set_coop(alg, 1); h.set_wave(alg, 1)
            lanes = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.inverse_dynamics_gradient_device(lanes.data_ptr(), d_in.data_ptr(), 3 * n, K)
            h.synchronize()
            assert h.get_coop(alg, K) == 0
            h.set_coop(alg, 0)
            assert h.get_coop(alg, K) == 2                  # automatic (the wave-per-configuration kernel is switched off above)
            outs = []
            for blocks in (0, 1, 2):
                out = torch.full((K + 2, 2 * n * n), 4.25, dtype=torch.float32, device="cuda")
                h.inverse_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks)
                h.synchronize()
                o = out.cpu().numpy()
                assert np.all(o[K:] == 4.25)
                outs.append(o[:K])
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
            err = relerr(outs[0], ref)[0]
            print("lean dID kernel K=%d: dc_du error %.2e (lane-per-configuration kernel %.2e)" % (K, err, relerr(lanes.cpu().numpy(), ref)[0]))
            assert err < TOL[robot]["dc_du"], (K, err)
            assert relerr(outs[0], lanes.cpu().numpy().astype(np.float64))[0] < 2 * TOL[robot]["dc_du"]
            zero = np.abs(ref).max(axis=0) == 0.0
            assert np.all(outs[0][:, zero] == 0.0)
            # compressed input rows (q, qd only: stride 2n) give the same bits
            d_c = torch.from_numpy(np.ascontiguousarray(np.concatenate([q, qd], axis=1), dtype=np.float32)).cuda()
            out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.inverse_dynamics_gradient_device(out.data_ptr(), d_c.data_ptr(), 2 * n, K)
            h.synchronize()
            assert np.array_equal(out.cpu().numpy(), outs[0])
        # a given qdd: not this kernel
        K = 200
        q, qd, u = make_inputs(n, K, 77)
        d_in = torch.from_numpy(pack(q, qd, u)).cuda()
        qdd = np.random.default_rng(78).uniform(-1, 1, (K, n)).astype(np.float32)
        d_qdd = torch.from_numpy(qdd).cuda()
        out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.inverse_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, d_qdd=d_qdd.data_ptr())
        h.synchronize()
        from oracle import rbd_oracle as O
        dc = O.rnea_grad(T, q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64))
        ref = np.concatenate([O.flat_colmajor(dc[:, :, :n]), O.flat_colmajor(dc[:, :, n:])], axis=1)
        assert relerr(out.cpu().numpy(), ref)[0] < TOL[robot]["dc_du"]
        h.set_wave(alg, 0)


@pytest.mark.gpu
def test_lean_inverse_dynamics_gradient_full_size_atlas30_65536(tables):
    """BASELINE config 4 (Atlas-30, batch 65536), the inverse-dynamics gradient through the automatic choice: spread sample against the
    oracle, every row finite, permutation of the batch permutes the rows bit for bit."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    with host.GridHandle("atlas30", device=0, precision=host.DEFAULT_PRECISION) as h:
        n, K, alg = h.n, 65536, host.ALG_ID_DU
        assert h.get_coop(alg, K) == 2
        q, qd, u = make_inputs(n, K, 5)
        x = pack(q, qd, u)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.inverse_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3 * n, K)
        h.synchronize()
        assert bool(torch.isfinite(d_out).all().item())
        rows = np.unique(np.concatenate([np.arange(64), np.linspace(0, K - 1, 96).astype(int), np.arange(K - 64, K)]))
        ref = oracle_all(tables("atlas30"), q[rows], qd[rows], u[rows])["dc_du_noqdd"]
        dc_rows = d_out[torch.from_numpy(rows).cuda()].cpu().numpy()
        assert relerr(dc_rows, ref)[0] < TOL["atlas30"]["dc_du"]
        perm = torch.from_numpy(np.random.default_rng(6).permutation(K)).cuda()
        d_out_p = torch.empty_like(d_out)
        d_in_p = d_in[perm].contiguous()
        h.inverse_dynamics_gradient_device(d_out_p.data_ptr(), d_in_p.data_ptr(), 3 * n, K)
        h.synchronize()
        assert bool(torch.equal(d_out_p, d_out[perm]))


@pytest.mark.gpu
def test_mixed_library_has_the_lean_inverse_dynamics_gradient_only(tables):
    """The mixed-arithmetic Atlas-30 library: its inverse-dynamics gradient has no double part, so the register-lean kernel is in it and
    automatic (same numbers as the fp32 library's, bit for bit); its forward-dynamics gradient keeps the 4-wave kernel (the lean block
    in the mixed arithmetic was measured and not shipped: profiles/r04/mixed_lean_report.txt)."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL_BY_PRECISION, oracle_all, pack
    T = tables("atlas30")
    with host.GridHandle("atlas30", device=0, precision="mixed") as hm, host.GridHandle("atlas30", device=0, precision="fp32") as hf:
        n, K = hm.n, 700
        assert hm.lean_available(host.ALG_ID_DU) and not hm.lean_available(host.ALG_FD_DU)
        hm.set_wave(host.ALG_ID_DU, 1); hf.set_wave(host.ALG_ID_DU, 1)
        assert hm.get_coop(host.ALG_ID_DU, K) == 2 and hm.get_coop(host.ALG_FD_DU, 16384) == 1
        q, qd, u = make_inputs(n, K, 123)
        d_in = torch.from_numpy(pack(q, qd, u)).cuda()
        outs = []
        for h in (hm, hf):
            o = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.inverse_dynamics_gradient_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
            outs.append(o.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
        assert relerr(outs[0], oracle_all(T, q, qd, u)["dc_du_noqdd"])[0] < TOL_BY_PRECISION["mixed"]["atlas30"]["dc_du"]


@pytest.mark.gpu
def test_lean_forward_dynamics_kernel_on_gpu(tables):
    """`forward_dynamics_kernel_coop8` through the C ABI: automatic for Atlas-30 above the wave-per-configuration kernel's batch sizes
    (grid_get_coop = 2), qdd within the north star's 1e-6 of the oracle and to round-off of the lane-per-configuration kernel, ragged
    batches, few blocks, rows past the batch untouched, no scratch."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import NORTH_STAR, TOL, oracle_all, pack
    robot, alg = "atlas30", host.ALG_FD
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        assert h.lean_available(alg)
        attrs = h.L.kernel_attributes(alg, coop=2)
        assert attrs["numRegs"] <= 256 and attrs["scratch_bytes_per_lane"] == 0, attrs
        n = h.n
        assert h.get_wave(alg, 64) and h.get_coop(alg, 4096) == 2          # small batches keep the wave-per-configuration kernel
        assert h.get_coop(alg, 32768) == 2 and h.get_coop(alg, 65536) == 0 and not h.get_wave(alg, 1024)    # ... large ones the lane kernel
        for K in (1, 70, 333, 1500):
            q, qd, u = make_inputs(n, K, 290 + K)
            ref = oracle_all(T, q, qd, u)["qdd"]
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            h.set_coop(alg, 1); h.set_wave(alg, 1)
            lanes = torch.zeros((K, n), dtype=torch.float32, device="cuda")
            h.forward_dynamics_device(lanes.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
            h.set_coop(alg, 3)
            outs = []
            for blocks in (0, 1, 2):
                out = torch.full((K + 2, n), 4.25, dtype=torch.float32, device="cuda")
                h.forward_dynamics_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks); h.synchronize()
                o = out.cpu().numpy()
                assert np.all(o[K:] == 4.25)
                outs.append(o[:K])
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
            err = relerr(outs[0], ref)[0]
            print("lean FD kernel K=%d: qdd error %.2e (lane-per-configuration kernel %.2e)" % (K, err, relerr(lanes.cpu().numpy(), ref)[0]))
            assert err < min(TOL[robot]["qdd"], NORTH_STAR), (K, err)
            assert relerr(outs[0], lanes.cpu().numpy().astype(np.float64))[0] < 2 * TOL[robot]["qdd"]
        h.set_coop(alg, 0); h.set_wave(alg, 0)


@pytest.mark.gpu
def test_lean_direct_minv_kernel_on_gpu(tables):
    """`direct_minv_kernel_coop8` through the C ABI: Minv against the oracle and the lane-per-configuration kernel, ragged batches, few
    blocks, rows past the batch untouched, an input of q only (stride NUM_JOINTS: the kernel must read nothing else), no scratch."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    robot, alg = "atlas30", host.ALG_MINV
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        assert h.lean_available(alg)
        attrs = h.L.kernel_attributes(alg, coop=2)
        assert attrs["numRegs"] <= 256 and attrs["scratch_bytes_per_lane"] == 0, attrs
        n = h.n
        assert h.get_coop(alg, 4096) == 2 and h.get_coop(alg, 16384) == 2 and h.get_coop(alg, 32768) == 0
        for K in (1, 70, 333, 1500):
            q, qd, u = make_inputs(n, K, 390 + K)
            ref = oracle_all(T, q, qd, u)["Minv"]
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            h.set_coop(alg, 1); h.set_wave(alg, 1)
            lanes = torch.zeros((K, n * n), dtype=torch.float32, device="cuda")
            h.direct_minv_device(lanes.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
            h.set_coop(alg, 3)
            outs = []
            for blocks in (0, 1, 2):
                out = torch.full((K + 2, n * n), 4.25, dtype=torch.float32, device="cuda")
                h.direct_minv_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks); h.synchronize()
                o = out.cpu().numpy()
                assert np.all(o[K:] == 4.25)
                outs.append(o[:K])
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
            err = relerr(outs[0], ref)[0]
            print("lean Minv kernel K=%d: error %.2e (lane-per-configuration kernel %.2e)" % (K, err, relerr(lanes.cpu().numpy(), ref)[0]))
            assert err < TOL[robot]["Minv"], (K, err)
            assert np.all(outs[0][:, np.abs(ref).max(axis=0) == 0.0] == 0.0)
            # q only, tightly packed, at the very end of its allocation: reading qd or u would run past it
            d_q = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).cuda()
            out = torch.zeros((K, n * n), dtype=torch.float32, device="cuda")
            h.direct_minv_device(out.data_ptr(), d_q.data_ptr(), n, K); h.synchronize()
            assert np.array_equal(out.cpu().numpy(), outs[0])
        h.set_coop(alg, 0); h.set_wave(alg, 0)
