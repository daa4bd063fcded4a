"""Tile-cooperative forward-dynamics-gradient kernel: one block of 4 wavefronts per tile of 64 configurations; one wave runs the
Minv recursion while the others run RNEA, results cross through LDS (Minv, c, qdd), every wave then differentiates its own
group of columns (gridcodegenerator_amd/emit/cores.py: core_forward_dynamics_gradient_coop / core_gradient_recompute(coop=...)).
The regime it replaces is the reference's block-per-configuration design (GRiDCodeGenerator.py:72-83,
algorithms/_forward_dynamics_gradient.py:7-57)."""
import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd.emit import cores
from gridcodegenerator_amd.emit.model import RobotSpec


def emulate_block(spec, builder, groups, q, qd, u, ksplit=None, park=0):
    """Interpret the cores of one block on the CPU: the exchange region is a dict that every core reads and writes; three
    sweeps reach the fixed point (Minv and c are published first, then qdd by the producer)."""
    n, K = spec.n, q.shape[0]
    slots = cores.CoopSlots(spec)
    slots.ksplit = ksplit
    if park:        # consumers run the d/dqd recursion of up to `park` of their columns ahead of the barriers
        slots.hoist_cost = [1] * n
        slots.hoist_budget = {"consumer": 10 ** 9, "consumer_c": 10 ** 9}
        slots.hoist_max_columns = park
    traces = [builder(role, cols, slots) for (role, cols) in groups]
    base = {"gravity": np.full(K, 9.81)}
    for j in range(n):
        base["in.q(%d)" % j] = q[:, j]; base["in.qd(%d)" % j] = qd[:, j]; base["in.u(%d)" % j] = u[:, j]
    xch = {"in.xch_get(%d)" % s: np.zeros(K) for s in range(slots.count)}
    got = np.zeros((K, 2 * n * n))
    for sweep in range(3):
        # c and qdd share slots: replay the block in program order -- phase 1 publishes (Minv | c), phase 2 (producer) qdd
        new = {}
        for tr, (role, cols) in zip(traces, groups):
            inp = dict(base); inp.update(xch)
            outs = tr.evaluate(inp)
            seen_barriers = 0
            for (dst, _), o in zip(tr.outputs, outs):
                if dst == "barrier":
                    seen_barriers += 1
                elif isinstance(dst, str) and dst.startswith("xch:"):
                    if (sweep == 0 and seen_barriers == 0) or (sweep >= 1 and seen_barriers == 1):
                        new["in.xch_get(%s)" % dst[4:]] = np.broadcast_to(o, (K,)).astype(np.float32).astype(np.float64)
                elif not isinstance(dst, str) and sweep == 2:
                    L = n * len(cols)
                    half, rem = divmod(int(dst), L)
                    ci, r = divmod(rem, n)
                    got[:, half * n * n + n * cols[ci] + r] = o
        xch.update(new)
    return got, traces


def test_two_producer_waves_match_oracle(robot_name, robots, tables):
    """Two producer waves (large robots): both run the backward pass of the Minv recursion (each only the part its columns need),
    each finishes its own columns of the forward pass and its share of qdd = Minv (u - c); every wave reads qdd as the sum."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(robot_name))
    n, K = spec.n, 4
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 23))
    ref = O.fd_grad(tables(robot_name), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    b = [0, n // 4, n // 2, 3 * n // 4, n]
    groups = [("producer", list(range(b[3], b[4]))), ("producer2", list(range(b[2], b[3]))), ("consumer_c", list(range(b[0], b[1]))),
              ("consumer", list(range(b[1], b[2])))]
    builder = lambda role, cols, sl: cores.core_gradient_recompute(spec, "fd", cols=cols, coop=(role, sl))
    got, traces = emulate_block(spec, builder, groups, q, qd, u, ksplit=(n + 1) // 2)
    assert relerr(got, ref)[0] < 5e-6
    slots = cores.CoopSlots(spec)
    published = []
    for tr, (role, cols) in zip(traces, groups):
        dsts = [d for (d, _) in tr.outputs]
        assert dsts.count("barrier") == 2
        published.append(set(int(d[4:]) for d in dsts if isinstance(d, str) and d.startswith("xch:")))
    minv_slots = set(slots.minv.values())
    assert published[0] & minv_slots and published[1] & minv_slots and not (published[0] & published[1])       # disjoint shares
    assert (published[0] | published[1]) >= minv_slots                                                        # that cover Minv
    assert set(slots.qdd) <= published[0] and set(slots.qdd2) <= published[1] and set(slots.c) <= published[2]
    # each producer's trace keeps only part of the recursion: fewer operations than the single producer's
    single = builder("producer", [], cores.CoopSlots(spec))
    s2 = cores.CoopSlots(spec); s2.ksplit = (n + 1) // 2
    assert all(cores._arith_ops(builder(r, [], s2)) < cores._arith_ops(single) for r in ("producer", "producer2"))


def test_parked_dqd_recursions_match_oracle(robot_name, robots, tables):
    """Consumer waves compute the d/dqd half of some of their columns BEFORE the barriers (it needs neither qdd nor Minv until its
    final product) and keep the values until the column's products: same results, work moved in front of the first barrier."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(robot_name))
    n, K = spec.n, 4
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 29))
    ref = O.fd_grad(tables(robot_name), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    b = [0, n // 4, n // 2, 3 * n // 4, n]
    groups = [("producer", list(range(b[3], b[4]))), ("producer2", list(range(b[2], b[3]))), ("consumer", list(range(b[0], b[1]))),
              ("consumer_c", list(range(b[1], b[2])))]
    builder = lambda role, cols, sl: cores.core_gradient_recompute(spec, "fd", cols=cols, coop=(role, sl))
    plain, tr0 = emulate_block(spec, builder, groups, q, qd, u, ksplit=(n + 1) // 2)
    got, tr1 = emulate_block(spec, builder, groups, q, qd, u, ksplit=(n + 1) // 2, park=2)
    assert relerr(got, ref)[0] < 5e-6
    assert np.abs(got - plain).max() <= 1e-6 * np.abs(ref).max()           # (same arithmetic, emitted earlier)

    def before_first_barrier(tr):
        live = tr.live_nodes()
        b0 = [pos for (dst, _), pos in zip(tr.outputs, tr.out_pos) if dst == "barrier"][0]
        return sum(1 for k in range(1, b0) if live[k] and tr.nodes[k][0] in ("fma", "mul", "add"))
    for t0, t1, (role, cols) in zip(tr0, tr1, groups):
        if role.startswith("consumer") and cols:
            assert before_first_barrier(t1) > before_first_barrier(t0) or role == "consumer"       # work moved ahead of the barriers
            assert [d for (d, _) in t1.outputs].count("barrier") == 2
        else:
            assert before_first_barrier(t1) == before_first_barrier(t0)


@pytest.mark.parametrize("schedule", ["fused", "recompute"])
def test_coop_cores_match_oracle(robot_name, schedule, robots, tables):
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(robot_name))
    n, K = spec.n, 4
    if schedule == "fused" and n > 12:
        pytest.skip("large robots use the recomputing schedule")
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 19))
    ref = O.fd_grad(tables(robot_name), q, qd, u)
    ref = np.concatenate([O.flat_colmajor(ref[:, :, :n]), O.flat_colmajor(ref[:, :, n:])], axis=1)
    b = [0, n // 4, n // 2, 3 * n // 4, n]
    groups = [("producer", list(range(b[3], b[4]))), ("consumer_c", list(range(b[0], b[1]))), ("consumer", list(range(b[1], b[2]))),
              ("consumer", list(range(b[2], b[3])))]
    if schedule == "fused":
        builder = lambda role, cols, sl: cores.core_forward_dynamics_gradient_coop(spec, role, cols, sl)
    else:
        builder = lambda role, cols, sl: cores.core_gradient_recompute(spec, "fd", cols=cols, coop=(role, sl))
    got, traces = emulate_block(spec, builder, groups, q, qd, u)
    # the exchange region holds fp32: Minv, c and qdd are rounded once on their way through it
    assert relerr(got, ref)[0] < 5e-6
    for tr, (role, cols) in zip(traces, groups):
        dsts = [d for (d, _) in tr.outputs]
        assert dsts.count("barrier") == 2                      # every wave executes the same two block barriers
        puts = [d for d in dsts if isinstance(d, str) and d.startswith("xch:")]
        assert (len(puts) > n) == (role == "producer")
    # the producer is the only core that contains the Minv recursion
    rcp = [sum(v for k, v in tr.op_counts().items() if k.startswith("rcp")) for tr in traces]
    assert rcp[0] > 0 and rcp[1:] == [0, 0, 0]


def test_generator_emits_coop_kernel(tmp_path, monkeypatch, robots):
    from gridcodegenerator_amd import GRiDCodeGenerator
    monkeypatch.chdir(tmp_path)
    gen = GRiDCodeGenerator(robots("mixed5"))
    gen.gen_all_code()
    code = gen.code_str
    assert "void forward_dynamics_gradient_kernel_coop(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u" in code
    assert "const int FD_DU_COOP_WAVES = 4;" in code and "in.barrier();" in code and "in.xch_put(" in code and "in.xch_get(" in code
    roles = [r for (r, c) in gen.coop_stats["groups"]]
    assert roles == ["producer", "consumer_c", "consumer", "consumer"]
    cols = sorted(c for (_, cs) in gen.coop_stats["groups"] for c in cs)
    assert cols == list(range(5))
    assert gen.coop_stats["lds_bytes"] <= 160 * 1024


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["iiwa7", "mixed5", "quad12", "atlas30"])
def test_coop_kernel_on_gpu(robot, tables):
    """Through the C ABI: the tile-cooperative kernel against the oracle and against the single-wave kernel, ragged batches,
    few blocks (grid-stride over tiles), rows past the batch untouched."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    host.build_library(robot, host.DEFAULT_PRECISION)
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        assert h.coop_available(host.ALG_FD_DU)
        n = h.n
        for K in (1, 70, 333):
            q, qd, u = make_inputs(n, K, 90 + K)
            ref = oracle_all(T, q, qd, u)
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            h.set_coop(host.ALG_FD_DU, 1); h.set_split(host.ALG_FD_DU, 1)
            plain = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.forward_dynamics_gradient_device(plain.data_ptr(), d_in.data_ptr(), 3 * n, K)
            h.synchronize()
            h.set_coop(host.ALG_FD_DU, 2); h.set_split(host.ALG_FD_DU, 0)
            assert h.get_coop(host.ALG_FD_DU, K)
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
            # (the tolerances are batch maxima of max|err| / max|ref|; over one or a few configurations the ratio is noisier)
            assert err < TOL[robot]["df_du"] * (4 if K < 64 else 1), (robot, K, err)
            assert relerr(outs[0], plain.cpu().numpy().astype(np.float64))[0] < 2 * TOL[robot]["df_du"]
        h.set_coop(host.ALG_FD_DU, 0)


@pytest.mark.gpu
def test_coop_kernel_full_size_iiwa7_16384(tables):
    """BASELINE.json configs[2] through the tile-cooperative kernel: spread sample against the oracle + permutation invariance."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    with host.GridHandle("iiwa7", device=0, precision=host.DEFAULT_PRECISION) as h:
        n, K = h.n, 16384
        h.set_coop(host.ALG_FD_DU, 2)
        q, qd, u = make_inputs(n, K, 3)
        x = pack(q, qd, u)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3 * n, K)
        h.synchronize()
        df = d_out.cpu().numpy()
        rows = np.unique(np.concatenate([np.arange(64), np.linspace(0, K - 1, 192).astype(int), np.arange(K - 64, K)]))
        ref = oracle_all(tables("iiwa7"), q[rows], qd[rows], u[rows])
        assert relerr(df[rows], ref["df_du"])[0] < TOL["iiwa7"]["df_du"]
        perm = np.random.default_rng(4).permutation(K)
        d_in_p = torch.from_numpy(np.ascontiguousarray(x[perm])).cuda()
        d_out_p = torch.empty_like(d_out)
        h.forward_dynamics_gradient_device(d_out_p.data_ptr(), d_in_p.data_ptr(), 3 * n, K)
        h.synchronize()
        assert np.array_equal(d_out_p.cpu().numpy(), df[perm])


def test_f_table_replaces_subtree_walks(robots):
    """Recomputing schedule, tile-cooperative cores: a wave parks the accumulated force of every column joint it finishes in its
    own exchange slots (CoopSlots.f) and runs its columns deepest first, so the force of a column's joint is its local part plus
    six LDS reads per child column instead of a walk over the whole subtree.  Same outputs (emulate_block tests above), less
    arithmetic after the second barrier; slots are written once, before they are read, and only by the wave that reads them."""
    spec = RobotSpec(robots("atlas30"))
    arith = ("fma", "mul", "add")

    def after_barriers(f_table, role, cols):
        slots = cores.CoopSlots(spec)
        slots.ksplit = 15
        slots.f_table = f_table
        tr = cores.core_gradient_recompute(spec, "fd", cols=cols, coop=(role, slots))
        live = tr.live_nodes()
        b = [pos for (dst, _), pos in zip(tr.outputs, tr.out_pos) if dst == "barrier"]
        puts = [int(d[4:]) for (d, _) in tr.outputs if isinstance(d, str) and d.startswith("tab:")]
        return sum(1 for k in range(b[1], len(tr.nodes)) if live[k] and tr.nodes[k][0] in arith), puts, slots
    for role, cols in (("consumer", [3, 4, 5, 6, 7, 8, 9]), ("producer2", [10, 11, 12, 13, 14, 15, 16])):
        with_table, puts, slots = after_barriers(True, role, cols)
        without, no_puts, _ = after_barriers(False, role, cols)
        assert no_puts == [] and with_table < 0.93 * without, (role, with_table, without)
        assert sorted(puts) == sorted(slots.f[j] + r for j in cols for r in range(6))       # every column joint, once
    assert 4 * (4 * 64 * 32 + 64 * cores.CoopSlots(spec).count) <= 160 * 1024                # staging + exchange region fit the CU's LDS
