"""The oracle (oracle/rbd_oracle.py) against the fixtures produced by the reference itself
(tests/golden/make_golden.py), plus reference-independent identities."""
import numpy as np
import pytest

from oracle import rbd_oracle as O

TOL = 1e-11  # float64 restatement vs float64 reference: agreement to round-off


def _err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def test_rnea_matches_reference(robot_name, golden, tables):
    G, T = golden(robot_name), tables(robot_name)
    c, v, a, f = O.rnea(T, G["q"], G["qd"])
    assert _err(c, G["c_noqdd"]) < TOL
    c, v, a, f = O.rnea(T, G["q"], G["qd"], G["qdd"])
    assert _err(c, G["c_qdd"]) < TOL
    assert _err(v, G["v"]) < TOL and _err(a, G["a"]) < TOL and _err(f, G["f"]) < TOL


def test_minv_matches_reference(robot_name, golden, tables):
    G, T = golden(robot_name), tables(robot_name)
    assert _err(O.minv(T, G["q"], False), G["Minv_upper"]) < TOL
    assert _err(O.minv(T, G["q"], True), G["Minv_dense"]) < TOL
    Minv, F, U, Dinv = O.minv_bpass(T, G["q"])
    assert _err(U, G["U"]) < TOL and _err(Dinv, G["Dinv"]) < TOL
    assert np.all(np.tril(O.minv(T, G["q"], False), -1) == 0.0)


def test_forward_dynamics_matches_reference(robot_name, golden, tables):
    G, T = golden(robot_name), tables(robot_name)
    assert _err(O.forward_dynamics(T, G["q"], G["qd"], G["u"]), G["qdd"]) < TOL


def test_gradients_match_reference(robot_name, golden, tables):
    """prismatic_fix=False is the faithful restatement and must match the reference on every robot."""
    G, T = golden(robot_name), tables(robot_name)
    assert _err(O.rnea_grad(T, G["q"], G["qd"], None, prismatic_fix=False), G["dc_du_noqdd"]) < TOL
    assert _err(O.rnea_grad(T, G["q"], G["qd"], G["qdd"], prismatic_fix=False), G["dc_du_qdd"]) < TOL
    assert _err(O.fd_grad(T, G["q"], G["qd"], G["u"], prismatic_fix=False), G["df_du"]) < TOL


@pytest.mark.parametrize("name", ["iiwa7", "atlas30"])
def test_prismatic_fix_is_noop_for_revolute_robots(name, golden, tables):
    G, T = golden(name), tables(name)
    a = O.fd_grad(T, G["q"], G["qd"], G["u"], prismatic_fix=True)
    b = O.fd_grad(T, G["q"], G["qd"], G["u"], prismatic_fix=False)
    assert np.array_equal(a, b)


def _fd_numeric(T, q, qd, u, eps=1e-6):
    K, n = q.shape
    num = np.zeros((K, n, 2 * n))
    for i in range(n):
        e = np.zeros(n); e[i] = eps
        num[:, :, i] = (O.forward_dynamics(T, q + e, qd, u) - O.forward_dynamics(T, q - e, qd, u)) / (2 * eps)
        num[:, :, n + i] = (O.forward_dynamics(T, q, qd + e, u) - O.forward_dynamics(T, q, qd - e, u)) / (2 * eps)
    return num


def test_fd_grad_matches_finite_differences(robot_name, golden, tables):
    G, T = golden(robot_name), tables(robot_name)
    q, qd, u = G["q"][:3], G["qd"][:3], G["u"][:3]
    num = _fd_numeric(T, q, qd, u)
    ana = O.fd_grad(T, q, qd, u, prismatic_fix=True)
    assert np.abs(num - ana).max() / np.abs(ana).max() < 2e-7


def test_reference_gradient_is_wrong_for_prismatic_joints(golden, tables):
    """Documents the reference defect (_test.py:311,437): motion cross product applied to a force."""
    G, T = golden("mixed5"), tables("mixed5")
    q, qd, u = G["q"][:3], G["qd"][:3], G["u"][:3]
    num = _fd_numeric(T, q, qd, u)
    ref_mode = O.fd_grad(T, q, qd, u, prismatic_fix=False)
    assert np.abs(num - ref_mode).max() / np.abs(num).max() > 1e-2


def test_minv_times_mass_matrix_is_identity(robot_name, golden, tables):
    G, T = golden(robot_name), tables(robot_name)
    q = G["q"][:4]
    K, n = q.shape
    M = np.zeros((K, n, n))
    for i in range(n):
        e = np.zeros((K, n)); e[:, i] = 1.0
        M[:, :, i] = O.rnea(T, q, np.zeros((K, n)), e, gravity=0.0)[0]
    Mi = O.minv(T, q, True)
    assert np.abs(np.einsum("kij,kjl->kil", Mi, M) - np.eye(n)).max() < 1e-10


def test_single_configuration_plumbing(golden, tables):
    """BASELINE.json configs[0]: iiwa-7 RNEA, batch 1, on the CPU."""
    G, T = golden("iiwa7"), tables("iiwa7")
    c = O.rnea(T, G["q"][0], G["qd"][0])[0]
    assert c.shape == (1, 7) and _err(c[0], G["c_noqdd"][0]) < TOL
