"""Register-lean tile-cooperative forward-dynamics-gradient kernel (`forward_dynamics_gradient_kernel_coop8`): EIGHT wavefronts per
tile of 64 configurations -- two per SIMD, at most 256 registers each -- instead of four that own a SIMD and 464 registers.  What
makes the cores lean (emit/cores.py: CoopSlots.enable_lean, LeanRole, lean_plan; emit/algorithms.py: direct_minv_lean): a
block-shared input table, the Minv recursion by base-rooted tree with U, 1/D and the backward-pass entries parked in LDS, bias
torques depth first, qdd rows after the barrier, X rebuilt for the way back up the tree, and gradient HALF columns as the unit of
work.  The block's cores are interpreted here on the CPU (numpy over the tracer IR) with the exchange region as a dictionary.
Reference mapping being replaced: algorithms/_forward_dynamics_gradient.py:7-57, algorithms/_inverse_dynamics_gradient.py:199-246,501-540."""
import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd.emit import cores
from gridcodegenerator_amd.emit.model import RobotSpec


def emulate_lean_block(spec, slots, plan, q, qd, u):
    """The block's barriers separate its phases; phase p of every wave reads only what phases < p published.  Sweep p evaluates every
    wave's core against the exchange region as it stands and publishes the `xch` writes of PHASE p only (a Minv slot is written
    twice -- the backward pass's value before B1, the final value between B1 and B2 -- and the forward pass must read the former);
    the outputs are taken from the last sweep."""
    n, K = spec.n, q.shape[0]
    traces = [cores.core_gradient_recompute(spec, "fd", cols=items, coop=(role, slots)) for (role, items) in plan]
    base = {"gravity": np.full(K, 9.81)}
    for j in range(n):
        base["in.q(%d)" % j] = q[:, j]; base["in.qd(%d)" % j] = qd[:, j]; base["in.u(%d)" % j] = u[:, j]
    xch = {"in.xch_get(%d)" % s: np.zeros(K) for s in range(-7 * n, slots.count)}      # (negative: the parking words below the region)
    got = np.full((K, 2 * n * n), np.nan)
    phases = cores.LEAN_BARRIERS + 1
    with np.errstate(all="ignore"):
        for sweep in range(phases):
            new = {}
            for tr in traces:
                inp = dict(base); inp.update(xch)
                outs = tr.evaluate(inp)
                phase = 0
                for (dst, _), o in zip(tr.outputs, outs):
                    if dst == "barrier":
                        phase += 1
                    elif isinstance(dst, str) and dst.startswith("xch:"):
                        if phase == sweep:
                            new["in.xch_get(%s)" % dst[4:]] = np.broadcast_to(o, (K,)).astype(np.float32).astype(np.float64)
                    elif not isinstance(dst, str) and sweep == phases - 1:
                        run, r = divmod(int(dst), n)
                        got[:, tr.run_bases[run] + r] = o
            xch.update(new)
    return got, traces


@pytest.mark.parametrize("robot", ["atlas30", "mixed5", "iiwa7"])
def test_lean_block_matches_oracle(robot, robots, tables):
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(robot))
    n, K = spec.n, 3
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 61))
    ref = O.fd_grad(tables(robot), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec)
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
    table = set(slots.itab["qd"]) | set(slots.itab["u"])
    for j in range(n):
        table |= {slots.itab["s"][j], slots.itab["c"][j]} if spec.uses_trig[j] else {slots.itab["q"][j]}
    assert table <= set(writers) and set(slots.c) <= set(writers) and set(slots.qdd) <= set(writers) and set(slots.minv.values()) <= set(writers)


def test_lean_block_with_per_column_minv_matches_oracle(robots, tables):
    """lean_plan(columns_from_chain=True): only the articulated-inertia chain (U, 1/D) of the Minv recursion is serial and published;
    the backward pass's F recursions move into the per-column phase (alg.minv_columns_lean) that all eight waves share."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots("atlas30"))
    n, K = spec.n, 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 62))
    ref = O.fd_grad(tables("atlas30"), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    slots, plan = cores.lean_plan(spec, columns_from_chain=True)
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
    slots, plan = cores.lean_plan(spec, order="runs")
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
    assert max(slots.lean_model["post"]) <= 1.05 * max(cores.lean_plan(spec)[0].lean_model["post"])     # balanced nearly as well as the scattered sets


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
    assert 4 * 64 * (slots.count + cores.LEAN_WAVES * spec.n) <= 160 * 1024
    assert 7 * spec.n <= cores.LEAN_WAVES * spec.n                      # U / 1/D parking fits the staging regions below the exchange region


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
        assert attrs["numRegs"] <= 256 and attrs["maxThreadsPerBlock"] >= 512, attrs
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
