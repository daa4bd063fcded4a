"""Wave-per-configuration forward-dynamics-gradient kernel (emit/wave.py): the lanes of ONE wavefront share a configuration
(reference mapping: one block per configuration, helpers/_code_generation_helpers.py:41-55).

CPU: the traced wave cores interpreted with numpy, the batch axis of the interpreter being the 64 LANES of one wave (broadcasts,
the uniform LDS table and the published Minv are interpreted as what they are), against the oracle; generation-time structure.
GPU (-m gpu): the kernel through the C ABI at K = 1, 7, 64, 200 against the oracle and against the lane-per-configuration kernel."""
import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd.emit import wave
from gridcodegenerator_amd.emit.model import RobotSpec, SubForest, base_trees
from gridcodegenerator_amd.robots import get_robot


def _lane_inputs(spec, first, m, q, qd, u, gravity, sfx=""):
    lanes = np.arange(wave.WAVE)
    kcol = lanes % m
    inputs = {"in.lane_q%s()" % sfx: q[first + kcol], "in.lane_qd%s()" % sfx: qd[first + kcol], "__kcol%s__" % sfx: kcol}
    if not sfx:
        inputs["in.lane_u()"] = u[first + kcol]
        inputs["gravity"] = np.full(wave.WAVE, gravity)
    upper = [(r, c) for r in range(6) for c in range(r, 6)]
    for e, (r, c) in enumerate(upper):
        inputs["in.lane_I%s(%d)" % (sfx, e)] = np.array([spec.Imats[first + k][r, c] for k in kcol])
    return inputs


def interpret_wave_cores(spec, q, qd, u, gravity=9.81, dtype="float64", use_roles=True, kind="fd_du", qdd=None):
    """df_du (n x 2n) of ONE configuration from the wave cores of every joint group, the helper / helped roles included: the
    helper's trace is interpreted first and what it wrote into the other wave's table is handed to that wave's trace.  Other kinds:
    "id" -> c (n), "minv" -> upper-triangular Minv (n x n), "fd" -> qdd (n), "id_du" -> dc_du (n x 2n)."""
    n = spec.n
    out = np.zeros({"id": (n,), "fd": (n,), "minv": (n, n)}.get(kind, (n, 2 * n)))
    use_roles = use_roles and kind == "fd_du"
    lanes = np.arange(wave.WAVE)
    groups = wave.wave_groups(spec)
    roles = wave.wave_roles(spec, groups) if use_roles else {}
    helped_by = {hd: hr for hr, hd in roles.items()}
    table_init = {}
    order = sorted(range(len(groups)), key=lambda w: 0 if w in roles else 1)          # helper waves first
    for w in order:
        first, m = groups[w]
        kw = {}
        inputs = _lane_inputs(spec, first, m, q, qd, u, gravity)
        if w in roles:
            f2, m2 = groups[roles[w]]
            kw["helper_for"] = SubForest(spec, f2, m2)
            inputs.update(_lane_inputs(spec, f2, m2, q, qd, u, gravity, "2"))
        if w in helped_by:
            kw["helped"] = True
            inputs["__utab_init__"] = table_init[w]
        tr = wave.core_forward_dynamics_gradient_wave(SubForest(spec, first, m), barriers=bool(roles), kind=kind, **kw)
        kcol = lanes % m
        inputs["in.lane_qdd()"] = (qdd[first + kcol] if qdd is not None else np.zeros(wave.WAVE))
        for node in tr.nodes[1:]:
            if node[0] != "in":
                continue
            expr = node[1]
            call = expr.split("/*")[0]
            if call.startswith("in.mask_k("):
                inputs[expr] = (kcol == int(call[len("in.mask_k("):-1])).astype(float)
            elif call.startswith("in.mask_dq("):
                inputs[expr] = (lanes == int(call[len("in.mask_dq("):-1])).astype(float)
            elif call.startswith("in.mask_dqd("):
                inputs[expr] = (lanes == m + int(call[len("in.mask_dqd("):-1])).astype(float)
            elif call.startswith("in.mask_row_le("):
                inputs[expr] = (int(call[len("in.mask_row_le("):-1]) <= kcol).astype(float)
        vals = tr.evaluate(inputs, dtype=dtype)
        assert sum(1 for (dst, _) in tr.outputs if dst == "barrier") == (1 if roles else 0)      # every wave of a block: the same count
        for (dst, _), val in zip(tr.outputs, vals):
            if isinstance(dst, int):
                val = np.asarray(val) + np.zeros(wave.WAVE)
                if kind == "id":
                    out[first + dst] = val[0]                              # a wave-uniform value per joint
                elif kind == "fd":
                    out[first:first + m] = val[:m]                         # lane k: qdd_k
                elif kind == "minv":
                    out[first + dst, first:first + m] = val[:m]            # lane k: column k
                else:
                    for lane in range(2 * m):
                        out[first + dst, first + (lane % m) + (n if lane >= m else 0)] = val[lane]
            elif isinstance(dst, str) and dst.startswith("utab2:"):
                table_init.setdefault(roles[w], {})[int(dst[6:])] = float(np.asarray(val).reshape(-1)[0])
    return out


@pytest.mark.parametrize("name", ["iiwa7", "mixed5", "quad12", "atlas30"])
def test_wave_cores_match_the_oracle_on_the_cpu(name, robots, tables):
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(name))
    n = spec.n
    q, qd, u = (a[0].astype(np.float64) for a in make_inputs(n, 1, 31))
    ref = O.fd_grad(tables(name), q[None], qd[None], u[None])[0]
    got = interpret_wave_cores(spec, q, qd, u)
    assert np.abs(got - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    got32 = interpret_wave_cores(spec, q, qd, u, dtype="float32")       # fp32 storage, fused multiply-add: what the kernel computes
    assert np.abs(got32 - ref).max() < 2e-4 * np.abs(ref).max()
    plain = interpret_wave_cores(spec, q, qd, u, use_roles=False)       # every wave on its own (no helper): the same numbers
    assert np.abs(plain - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["iiwa7", "mixed5", "quad12", "atlas30"])
def test_wave_cores_of_the_other_algorithms_on_the_cpu(name, robots, tables):
    """The same phases serve RNEA, Minv, forward dynamics and the RNEA gradient (the small-batch path of every kernel)."""
    from oracle import rbd_oracle as O
    spec = RobotSpec(robots(name))
    n = spec.n
    q, qd, u = (a[0].astype(np.float64) for a in make_inputs(n, 1, 32))
    qdd = 0.7 * u
    T = tables(name)
    close = lambda got, ref: np.abs(got - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert close(interpret_wave_cores(spec, q, qd, u, kind="id"), O.rnea(T, q[None], qd[None])[0][0])
    assert close(interpret_wave_cores(spec, q, qd, u, kind="id", qdd=qdd), O.rnea(T, q[None], qd[None], qdd[None])[0][0])
    assert close(interpret_wave_cores(spec, q, qd, u, kind="minv"), np.triu(O.minv(T, q[None])[0]))
    assert close(interpret_wave_cores(spec, q, qd, u, kind="fd"), O.forward_dynamics(T, q[None], qd[None], u[None])[0])
    assert close(interpret_wave_cores(spec, q, qd, u, kind="id_du"), O.rnea_grad(T, q[None], qd[None], None)[0])
    assert close(interpret_wave_cores(spec, q, qd, u, kind="id_du", qdd=qdd), O.rnea_grad(T, q[None], qd[None], qdd[None])[0])


def test_wave_groups_are_runs_of_base_trees(robots):
    """Atlas-30: torso + arms + neck | both legs; a single chain: one group; every joint in exactly one group; lanes suffice."""
    spec = RobotSpec(robots("atlas30"))
    assert base_trees(spec) == [(0, 18), (18, 6), (24, 6)]
    assert wave.wave_groups(spec) == [(0, 18), (18, 12)]
    assert wave.wave_groups(RobotSpec(robots("iiwa7"))) == [(0, 7)]
    assert wave.wave_roles(spec, wave.wave_groups(spec)) == {1: 0}        # the legs' wave runs the torso group's first RNEA pass
    assert wave.wave_roles(RobotSpec(robots("iiwa7")), [(0, 7)]) == {}
    for name in ("iiwa7", "mixed5", "quad12", "atlas30"):
        s = RobotSpec(robots(name))
        groups = wave.wave_groups(s)
        assert sorted(j for (f, m) in groups for j in range(f, f + m)) == list(range(s.n))
        assert all(2 * m <= wave.WAVE for (_, m) in groups)
        for (f, m) in groups:
            sub = SubForest(s, f, m)
            assert all(p == -1 or 0 <= p < m for p in sub.parent)


def test_generated_header_has_the_wave_kernel(tmp_path, monkeypatch, robots):
    from gridcodegenerator_amd.GRiDCodeGenerator import GRiDCodeGenerator
    monkeypatch.chdir(tmp_path)
    gen = GRiDCodeGenerator(robots("mixed5"), FILE_NAMESPACE="grid_w")
    gen.gen_all_code()
    code = gen.code_str
    assert "void forward_dynamics_gradient_kernel_wave(" in code and "grid_lanes::bcast(" in code
    assert "in.utab_put(" in code and "in.m_get(" in code and "const int FD_DU_WAVE_WAVES = 2;" in code
    assert "const int FD_DU_WAVE_AUTO_MAX_K = 0;" in code                   # small robots: on request only
    assert gen.wave_stats["groups"] == [(0, 4), (4, 1)]
    assert any("forward_dynamics_gradient_kernel_wave" in k for k in gen.kernel_instances)


# Norm-wise errors of the wave-per-configuration kernels MEASURED on MI355X at the batch sizes used below (worst of six seeds per cell,
# the tests' own seeds among them; tests/gpu_checks/wave_small_batch_errors.py -> profiles/r04/wave_small_batch_errors.txt, fp32
# arithmetic).  Every bound below is 3x its cell -- the rule of the whole suite -- instead of a blanket factor for small batches: over one
# or a few configurations max|err| / max|ref| is noisier, but only the forward-dynamics gradient shows it (Atlas-30: 9.9e-6 at K = 1
# against 3.7e-6 at K = 200).
WAVE_MEASURED = {
    "iiwa7": {
        "c": {1: 1.4e-07, 7: 2.1e-07, 64: 2.4e-07, 200: 2.6e-07},
        "c_qdd": {1: 1.9e-07, 7: 2.5e-07, 64: 3.4e-07, 200: 2.6e-07},
        "Minv": {1: 5.4e-08, 7: 5.5e-08, 64: 6.2e-08, 200: 6.7e-08},
        "qdd": {1: 8.9e-08, 7: 1.4e-07, 64: 8.4e-08, 200: 1.2e-07},
        "dc_du_noqdd": {1: 2.4e-07, 7: 1.9e-07, 64: 2.6e-07, 200: 2.3e-07},
        "dc_du": {1: 2e-07, 7: 2.1e-07, 64: 2.6e-07, 200: 2.5e-07},
        "df_du": {1: 6.1e-07, 7: 1.3e-06, 64: 8.3e-07, 200: 5.7e-07},
    },
    "mixed5": {
        "c": {1: 3.2e-07, 7: 4.6e-07, 64: 4.2e-07, 200: 3.8e-07},
        "c_qdd": {1: 3.9e-07, 7: 1.8e-07, 64: 2.8e-07, 200: 2.8e-07},
        "Minv": {1: 2.2e-08, 7: 3.7e-08, 64: 4e-08, 200: 4.2e-08},
        "qdd": {1: 8e-08, 7: 7.6e-08, 64: 1.7e-07, 200: 1.7e-07},
        "dc_du_noqdd": {1: 2.6e-07, 7: 2.3e-07, 64: 1.7e-07, 200: 2e-07},
        "dc_du": {1: 1.8e-07, 7: 1.9e-07, 64: 1.5e-07, 200: 2.2e-07},
        "df_du": {1: 5.1e-06, 7: 3.2e-06, 64: 3.3e-06, 200: 1.4e-06},
    },
    "quad12": {
        "c": {1: 1.2e-07, 7: 1.4e-07, 64: 1.5e-07, 200: 2.3e-07},
        "c_qdd": {1: 2e-07, 7: 1.5e-07, 64: 1.7e-07, 200: 1.9e-07},
        "Minv": {1: 7.6e-08, 7: 1.1e-07, 64: 1.5e-07, 200: 1.4e-07},
        "qdd": {1: 2e-07, 7: 1.7e-07, 64: 1.7e-07, 200: 2.5e-07},
        "dc_du_noqdd": {1: 1.5e-07, 7: 1.6e-07, 64: 1.6e-07, 200: 1.9e-07},
        "dc_du": {1: 1.5e-07, 7: 1.6e-07, 64: 1.7e-07, 200: 1.9e-07},
        "df_du": {1: 3e-07, 7: 2.7e-07, 64: 2.7e-07, 200: 2.8e-07},
    },
    "atlas30": {
        "c": {1: 9.1e-08, 7: 1.6e-07, 64: 2.6e-07, 200: 2.2e-07},
        "c_qdd": {1: 1.2e-07, 7: 2.1e-07, 64: 2.4e-07, 200: 2.8e-07},
        "Minv": {1: 1.1e-07, 7: 1.1e-07, 64: 1.6e-07, 200: 1.4e-07},
        "qdd": {1: 3.2e-07, 7: 1.7e-07, 64: 2.2e-07, 200: 2.8e-07},
        "dc_du_noqdd": {1: 1.9e-07, 7: 3.8e-07, 64: 3e-07, 200: 2.9e-07},
        "dc_du": {1: 5.6e-07, 7: 2.7e-07, 64: 2.5e-07, 200: 3.2e-07},
        "df_du": {1: 9.9e-06, 7: 9.7e-06, 64: 4.6e-06, 200: 3.7e-06},
    },
}


def wave_tol(robot, key, K):
    return 3.0 * WAVE_MEASURED[robot][key][K]


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["iiwa7", "mixed5", "quad12", "atlas30"])
def test_wave_kernel_on_gpu(robot, tables):
    """Through the C ABI at K = 1, 7, 64, 200 (+ a strided launch with few blocks): against the oracle, against the
    lane-per-configuration kernel, rows past the batch untouched; the automatic choice follows the generated header."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    host.build_library(robot, host.DEFAULT_PRECISION)
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        assert h.wave_available(host.ALG_FD_DU)
        n = h.n
        assert h.get_wave(host.ALG_FD_DU, 64) == (n > 12)                # automatic for large robots at small batches only
        assert not h.get_wave(host.ALG_FD_DU, 16384)
        for K in (1, 7, 64, 200):
            q, qd, u = make_inputs(n, K, 40 + K)
            ref = oracle_all(T, q, qd, u)
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            h.set_wave(host.ALG_FD_DU, 1); h.set_coop(host.ALG_FD_DU, 1); h.set_split(host.ALG_FD_DU, 1)
            plain = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.forward_dynamics_gradient_device(plain.data_ptr(), d_in.data_ptr(), 3 * n, K)
            h.synchronize()
            h.set_wave(host.ALG_FD_DU, 2); h.set_coop(host.ALG_FD_DU, 0); h.set_split(host.ALG_FD_DU, 0)
            assert h.get_wave(host.ALG_FD_DU, K) and not h.get_coop(host.ALG_FD_DU, K)
            outs = []
            for blocks in (0, 1, 3):
                out = torch.full((K + 2, 2 * n * n), 4.25, dtype=torch.float32, device="cuda")
                h.forward_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks)
                h.synchronize()
                o = out.cpu().numpy()
                assert np.all(o[K:] == 4.25)
                outs.append(o[:K])
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
            err = relerr(outs[0], ref["df_du"])[0]
            assert err < wave_tol(robot, "df_du", K), (robot, K, err)
            assert relerr(outs[0], plain.cpu().numpy().astype(np.float64))[0] < 2 * max(TOL[robot]["df_du"], wave_tol(robot, "df_du", K))
            # structural zeros (columns and rows of different base-rooted trees) are written as exact zeros
            zero = np.abs(ref["df_du"]).max(axis=0) == 0.0
            assert np.all(outs[0][:, zero] == 0.0)
        h.set_wave(host.ALG_FD_DU, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["iiwa7", "mixed5", "quad12", "atlas30"])
def test_wave_kernels_of_the_other_algorithms_on_gpu(robot, tables):
    """RNEA (with and without qdd), Minv, forward dynamics and the RNEA gradient (with and without qdd) through the C ABI with the
    wave-per-configuration kernels forced, K = 1, 7, 64, 200: against the oracle and against the lane-per-configuration kernels."""
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    from oracle import rbd_oracle as O
    host.build_library(robot, host.DEFAULT_PRECISION)
    T = tables(robot)
    algs = (host.ALG_ID, host.ALG_MINV, host.ALG_FD, host.ALG_ID_DU)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        n = h.n
        for K in (1, 7, 64, 200):
            q, qd, u = make_inputs(n, K, 60 + K)
            ref = oracle_all(T, q, qd, u)
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            qdd_alt = np.random.default_rng(70 + K).uniform(-1.0, 1.0, (K, n)).astype(np.float32)       # (not FD's qdd: RNEA(q, qd, FD(u)) = u cancels)
            d_qdd = torch.from_numpy(qdd_alt).cuda()
            q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd_alt))
            ref["c_qdd"] = O.rnea(T, q64, qd64, qdd64)[0]
            dc = O.rnea_grad(T, q64, qd64, qdd64)
            ref["dc_du"] = np.concatenate([O.flat_colmajor(dc[:, :, :n]), O.flat_colmajor(dc[:, :, n:])], axis=1)

            def run_all(blocks=0):
                res = {}
                mk = lambda cols: torch.full((K + 2, cols), 4.25, dtype=torch.float32, device="cuda")
                o = mk(n); h.inverse_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks); res["c"] = o
                o = mk(n); h.inverse_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, d_qdd=d_qdd.data_ptr(), blocks=blocks); res["c_qdd"] = o
                o = mk(n * n); h.direct_minv_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks); res["Minv"] = o
                o = mk(n); h.forward_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks); res["qdd"] = o
                o = mk(2 * n * n); h.inverse_dynamics_gradient_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks); res["dc_du_noqdd"] = o
                o = mk(2 * n * n); h.inverse_dynamics_gradient_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, d_qdd=d_qdd.data_ptr(), blocks=blocks); res["dc_du"] = o
                h.synchronize()
                out = {}
                for k, v in res.items():
                    v = v.cpu().numpy()
                    assert np.all(v[K:] == 4.25), k
                    out[k] = v[:K]
                return out

            for a in algs:
                h.set_wave(a, 1)
            plain = run_all()
            for a in algs:
                assert h.wave_available(a)
                h.set_wave(a, 2)
                assert h.get_wave(a, K)
            got = run_all()
            few = run_all(blocks=3)
            for a in algs:
                h.set_wave(a, 0)
            tol = TOL[robot]
            for key in got:
                assert np.array_equal(got[key], few[key]), key
                t = wave_tol(robot, key, K)
                assert relerr(got[key], ref[key])[0] < t, (robot, K, key, relerr(got[key], ref[key])[0])
                assert relerr(got[key], plain[key].astype(np.float64))[0] < 2 * t, (robot, K, key)
            # entries that couple different base-rooted trees are written as exact zeros
            spec = RobotSpec(get_robot(robot))
            tree = np.zeros(n, dtype=int)
            for t_id, (first, m) in enumerate(base_trees(spec)):
                tree[first:first + m] = t_id
            cross = (tree[:, None] != tree[None, :]).T.reshape(-1)         # column-major (col, row)
            assert np.all(got["dc_du"][:, np.concatenate([cross, cross])] == 0.0)
            assert np.all(got["Minv"][:, cross] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["iiwa7", "atlas30"])
def test_automatic_choice_follows_the_header_thresholds(robot, tables):
    """The C ABI picks the wave-per-configuration kernel of an algorithm exactly up to <ALG>_WAVE_AUTO_MAX_K of the generated header
    (0: never by itself), an explicit choice of another variant wins, and a batch AT the threshold computed through the automatic
    path agrees with the oracle (forward dynamics and its gradient: the two algorithms with a threshold for large robots)."""
    import re
    import torch
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    host.build_library(robot, host.DEFAULT_PRECISION)
    with open(host.library_paths(robot, host.DEFAULT_PRECISION)["header"]) as fh:
        text = fh.read()
    names = {host.ALG_ID: "ID", host.ALG_MINV: "MINV", host.ALG_FD: "FD", host.ALG_ID_DU: "ID_DU", host.ALG_FD_DU: "FD_DU"}
    limit = {a: int(re.search(r"const int %s_WAVE_AUTO_MAX_K = (\d+);" % nm, text).group(1)) for a, nm in names.items()}
    for a, nm in names.items():         # ... capped where a register-lean tile-cooperative kernel takes over earlier (<ALG>_LEAN_WAVE_MAX_K)
        m = re.search(r"const int %s_LEAN_WAVE_MAX_K = (\d+);" % nm, text)
        if m and int(m.group(1)) > 0:
            limit[a] = min(limit[a], int(m.group(1)))
    T = tables(robot)
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        n = h.n
        for a, lim in limit.items():
            assert h.wave_available(a)
            if lim == 0:
                assert not h.get_wave(a, 1) and not h.get_wave(a, 64)
            else:
                assert h.get_wave(a, 1) and h.get_wave(a, lim) and not h.get_wave(a, lim + 1)
        assert limit[host.ALG_FD_DU] == (768 if n > 12 else 0)
        h.set_split(host.ALG_FD_DU, 1)                               # an explicit choice of another variant wins
        assert not h.get_wave(host.ALG_FD_DU, 64)
        h.set_split(host.ALG_FD_DU, 0)
        if limit[host.ALG_FD] > 0:
            # ... and so does an explicit launch shape: blocks x threads means blocks of `threads` CONFIGURATIONS (the reference's
            # <<<block_dimms, thread_dimms>>>), which only the lane-per-configuration kernels honour -- the automatic choice must not
            # turn 4 blocks of 64 threads into 4 configurations in flight.  Bitwise: automatic + shape == wave kernels off + shape,
            # automatic without a shape == wave kernel forced.
            Kx = min(limit[host.ALG_FD], 256)
            q, qd, u = make_inputs(n, Kx, 78)
            d_in = torch.from_numpy(pack(q, qd, u)).cuda()
            res = {}
            for name, mode, kw in (("auto_shape", 0, dict(blocks=4, threads=64)), ("never_shape", 1, dict(blocks=4, threads=64)),
                                   ("auto", 0, {}), ("always", 2, {})):
                h.set_wave(host.ALG_FD, mode)
                o = torch.full((Kx, n), 9.5, dtype=torch.float32, device="cuda")
                h.forward_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, Kx, **kw)
                h.synchronize()
                res[name] = o.cpu().numpy()
            h.set_wave(host.ALG_FD, 0)
            assert np.array_equal(res["auto_shape"], res["never_shape"]) and np.array_equal(res["auto"], res["always"])
            assert not np.array_equal(res["auto"], res["never_shape"])          # (different kernels round differently)
        K = max(limit[host.ALG_FD], limit[host.ALG_FD_DU], 64)
        q, qd, u = make_inputs(n, K, 77)
        ref = oracle_all(T, q, qd, u)
        d_in = torch.from_numpy(pack(q, qd, u)).cuda()
        qdd = torch.full((K, n), 9.5, dtype=torch.float32, device="cuda")
        df = torch.full((K, 2 * n * n), 9.5, dtype=torch.float32, device="cuda")
        h.forward_dynamics_device(qdd.data_ptr(), d_in.data_ptr(), 3 * n, K)
        h.forward_dynamics_gradient_device(df.data_ptr(), d_in.data_ptr(), 3 * n, K)
        h.synchronize()
        assert relerr(qdd.cpu().numpy(), ref["qdd"])[0] < TOL[robot]["qdd"]
        assert relerr(df.cpu().numpy(), ref["df_du"])[0] < TOL[robot]["df_du"]
