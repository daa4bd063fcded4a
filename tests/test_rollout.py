"""Rollout consumer of the forward-dynamics gradient: one lane integrates one trajectory with
semi-implicit Euler, q and qd stay on-chip, every step writes x+, A = dx+/dx and B = dx+/du.  The reference has no such
kernel (it ships the `_device` tier for this use: README.md:26-29, algorithms/_forward_dynamics_gradient.py:59-99); the
definition is oracle/rbd_oracle.py: rollout_step."""
import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd.emit import cores
from gridcodegenerator_amd.emit.model import RobotSpec


def _row(T, q, qd, u, dt):
    from oracle import rbd_oracle as O
    qn, qdn, A, B = O.rollout_step(T, q, qd, u, dt)
    return np.concatenate([qn, qdn, O.flat_colmajor(A), O.flat_colmajor(B)], axis=1)


@pytest.mark.parametrize("schedule", ["fused", "recompute"])
def test_rollout_step_core_matches_oracle(robot_name, schedule, robots, tables):
    """CPU: the traced step core (what the kernel executes per lane and step), interpreted in float64."""
    spec = RobotSpec(robots(robot_name))
    n, K, dt = spec.n, 4, 0.01
    if schedule == "fused":
        if n > 12:
            pytest.skip("large robots use the recomputing schedule")
        tr, bases = cores.core_rollout_step(spec)
    else:
        bases = []
        tr = cores.core_gradient_recompute(spec, "fd", rollout=bases)
    assert sorted(bases) == [2 * n * k for k in range(1 + 3 * n)] and bases[0] == 0
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, K, 13))
    inputs = {"gravity": np.full(K, 9.81), "in.dt()": np.full(K, dt)}
    for j in range(n):
        inputs["in.q(%d)" % j] = q[:, j]; inputs["in.qd(%d)" % j] = qd[:, j]; inputs["in.u(%d)" % j] = u[:, j]
    got = np.zeros((K, cores.rollout_row_count(spec)))
    for (dst, _), o in zip(tr.outputs, tr.evaluate(inputs)):
        k, i = divmod(int(dst), 2 * n)
        got[:, bases[k] + i] = o
    assert relerr(got, _row(tables(robot_name), q, qd, u, dt))[0] < 1e-12


def test_oracle_rollout_linearisation_is_the_jacobian(tables):
    """The oracle's A, B are the Jacobians of its own step map (finite differences)."""
    from oracle import rbd_oracle as O
    T = tables("iiwa7")
    n, dt, eps = 7, 0.01, 1e-6
    q, qd, u = (a.astype(np.float64) for a in make_inputs(n, 1, 5))
    u = O.rnea(T, q, qd)[0] + 0.1 * u
    qn, qdn, A, B = O.rollout_step(T, q, qd, u, dt)
    f = lambda q_, qd_, u_: np.concatenate(O.rollout_step(T, q_, qd_, u_, dt)[:2], axis=1)[0]
    x0 = f(q, qd, u)
    for i in range(n):
        e = np.zeros((1, n)); e[0, i] = eps
        assert np.abs((f(q + e, qd, u) - x0) / eps - A[0][:, i]).max() < 2e-4 * max(1.0, np.abs(A[0][:, i]).max())
        assert np.abs((f(q, qd + e, u) - x0) / eps - A[0][:, n + i]).max() < 2e-4 * max(1.0, np.abs(A[0][:, n + i]).max())
        assert np.abs((f(q, qd, u + e) - x0) / eps - B[0][:, i]).max() < 2e-4 * max(1.0, np.abs(B[0][:, i]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("robot,K,steps", [("iiwa7", 4096, 32), ("mixed5", 200, 8), ("atlas30", 330, 6)])
def test_rollout_kernel_on_gpu(robot, K, steps, tables):
    """-m gpu parity of forward_dynamics_gradient_rollout_kernel (through the C ABI): T = 32, K = 4096 for iiwa-7.
    Checked per step at the GPU's own states (x_{t+1}, A_t, B_t against the oracle's step from the GPU's x_t: tolerances
    of the forward-dynamics outputs they are built from) and end to end against the oracle's float64 rollout (looser: the
    fp32 state error compounds over the steps)."""
    import torch
    from gridcodegenerator_amd import host
    from oracle import rbd_oracle as O
    from test_gpu_parity import TOL
    tol = TOL[robot]
    host.build_library(robot, host.DEFAULT_PRECISION)
    T = tables(robot)
    dt = 0.005
    with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
        n, row = h.n, h.rollout_row_count()
        assert row == 2 * n * (1 + 3 * n)
        q0, qd0, noise = make_inputs(n, K, 71)
        rng = np.random.default_rng(72)
        # torques near gravity compensation of the initial state keep the trajectories gentle over the horizon
        u0 = O.rnea(T, q0.astype(np.float64), qd0.astype(np.float64))[0]
        u_traj = (u0[None] + 0.05 * rng.uniform(-1, 1, (steps, K, n))).astype(np.float32)
        x0 = np.ascontiguousarray(np.concatenate([q0, qd0], axis=1))
        d_x0 = torch.from_numpy(x0).cuda()
        d_u = torch.from_numpy(np.ascontiguousarray(u_traj)).cuda()
        guard = 3.5
        d_traj = torch.full((steps + 1, K, row), guard, dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_rollout_device(d_traj.data_ptr(), d_x0.data_ptr(), d_u.data_ptr(), K, steps, dt)
        h.synchronize()
        traj = d_traj.cpu().numpy()
        # a second launch shape (two waves per block, few blocks: grid-stride tile loop) must agree bit for bit
        d_traj2 = torch.zeros((steps, K, row), dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_rollout_device(d_traj2.data_ptr(), d_x0.data_ptr(), d_u.data_ptr(), K, steps, dt, blocks=3, threads=128)
        h.synchronize()
        assert np.array_equal(d_traj2.cpu().numpy(), traj[:steps])
    assert np.all(traj[steps] == guard) and np.isfinite(traj[:steps]).all()
    sample = np.unique(np.concatenate([np.arange(min(K, 70)), np.linspace(0, K - 1, 120).astype(int), np.arange(max(0, K - 66), K)]))
    worst = dict(x=0.0, A=0.0, B=0.0)
    x_prev = x0[sample].astype(np.float64)
    for t in range(steps):
        ref = _row(T, x_prev[:, :n], x_prev[:, n:], u_traj[t][sample].astype(np.float64), dt)
        got = traj[t][sample].astype(np.float64)
        ex = relerr(got[:, :2 * n], ref[:, :2 * n])[0]
        eA = relerr(got[:, 2 * n:2 * n + 4 * n * n], ref[:, 2 * n:2 * n + 4 * n * n])[0]
        eB = relerr(got[:, 2 * n + 4 * n * n:], ref[:, 2 * n + 4 * n * n:])[0]
        worst = dict(x=max(worst["x"], ex), A=max(worst["A"], eA), B=max(worst["B"], eB))
        assert ex < 2e-6, (t, ex)                        # x+ = x + dt (...): dominated by x itself
        assert eA < tol["df_du"] and eB < tol["Minv"], (t, eA, eB)
        x_prev = got[:, :2 * n]                          # teacher forcing: the next step starts from the GPU's state
    # end to end against the float64 rollout
    xs, As, Bs = O.rollout(T, q0[sample].astype(np.float64), qd0[sample].astype(np.float64), u_traj[:, sample].astype(np.float64), dt)
    e2e = relerr(traj[steps - 1][sample][:, :2 * n], xs[-1])[0]
    assert e2e < 1e-4, e2e
    print("rollout %s K=%d T=%d: per-step worst norm-wise x %.1e A %.1e B %.1e; final state vs float64 rollout %.1e"
          % (robot, K, steps, worst["x"], worst["A"], worst["B"], e2e))
