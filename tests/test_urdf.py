"""URDF loader / exporter (gridcodegenerator_amd/urdf.py): round trips of the built-in robots, fixed-joint merging,
negative joint axes, rotated inertial frames, unsupported input."""
import numpy as np
import pytest

from conftest import make_inputs
from gridcodegenerator_amd import urdf
from gridcodegenerator_amd.robots import get_robot
from oracle import rbd_oracle as O


def _same_robot(a, b, tol=1e-12):
    n = a.get_num_joints()
    assert b.get_num_joints() == n and a.get_parent_id_array() == b.get_parent_id_array()
    rng = np.random.default_rng(0)
    for j in range(n):
        assert a.get_S_ind_by_id(j) == b.get_S_ind_by_id(j)
        assert a.get_joint_by_id(j).get_name() == b.get_joint_by_id(j).get_name()
        assert abs(a.get_damping_by_id(j) - b.get_damping_by_id(j)) < tol
        for q in rng.uniform(-3, 3, 3):
            assert np.abs(a.get_Xmat_Func_by_id(j)(q) - b.get_Xmat_Func_by_id(j)(q)).max() < tol
        assert np.abs(a.get_Imat_by_id(j) - b.get_Imat_by_id(j)).max() < tol * max(1.0, np.abs(a.get_Imat_by_id(j)).max())


@pytest.mark.parametrize("name", ["iiwa7", "atlas30", "mixed5", "quad12"])
def test_round_trip(name, tmp_path):
    robot = get_robot(name)
    text = urdf.robot_to_urdf(robot)
    _same_robot(robot, urdf.load_urdf(text))
    path = tmp_path / (name + ".urdf")
    path.write_text(text)
    _same_robot(robot, urdf.URDFParser().parse(str(path)))        # reference README usage: URDFParser().parse(file)


CHAIN = """<robot name="t">
  <link name="base"/>
  <link name="l1"><inertial><origin xyz="0.1 0 0.05" rpy="0 0 0"/><mass value="2.0"/><inertia ixx="0.02" ixy="0" ixz="0.001" iyy="0.03" iyz="0" izz="0.01"/></inertial></link>
  <link name="tool"><inertial><origin xyz="0 0.02 0.1" rpy="0.3 -0.2 0.5"/><mass value="0.7"/><inertia ixx="0.004" ixy="0.0005" ixz="0" iyy="0.003" iyz="0" izz="0.002"/></inertial></link>
  <link name="l2"><inertial><origin xyz="0 0 0.2" rpy="0 0 0"/><mass value="1.5"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.005"/></inertial></link>
  <joint name="j1" type="revolute"><parent link="base"/><child link="l1"/><origin xyz="0 0 0.3" rpy="0 0 0.4"/><axis xyz="0 0 %(sign)s1"/><dynamics damping="0.2"/></joint>
  <joint name="weld" type="fixed"><parent link="l1"/><child link="tool"/><origin xyz="0.2 0 0.1" rpy="0 1.5707963267948966 0"/></joint>
  <joint name="j2" type="continuous"><parent link="tool"/><child link="l2"/><origin xyz="0 0.05 0.15" rpy="0.1 0 0"/><axis xyz="0 1 0"/></joint>
</robot>"""


def _torques(robot, q, qd, qdd):
    return O.rnea(O.RobotTables(robot), q, qd, qdd)[0]


def test_fixed_joint_is_merged_into_its_parent_body():
    robot = urdf.load_urdf(CHAIN % dict(sign=""))
    assert robot.get_num_joints() == 2 and robot.get_parent_id_array() == [-1, 0]
    assert [robot.get_joint_by_id(j).get_name() for j in range(2)] == ["j1", "j2"]
    # body 1 = l1 + tool: masses add, the first moment is the sum of the parts' first moments in the j1 frame
    I1 = robot.get_Imat_by_id(0)
    assert abs(I1[5, 5] - 2.7) < 1e-12
    R_weld = urdf.rpy_to_rotation((0.0, 1.5707963267948966, 0.0))
    com_tool = R_weld @ np.array([0.0, 0.02, 0.1]) + np.array([0.2, 0.0, 0.1])
    first_moment = 2.0 * np.array([0.1, 0.0, 0.05]) + 0.7 * com_tool
    mc = I1[:3, 3:]
    assert np.abs(np.array([mc[2, 1], mc[0, 2], mc[1, 0]]) - first_moment).max() < 1e-12
    # j2 hangs off the tool link: its tree transform is weld o origin(j2)
    jt = robot.get_joint_by_id(1)
    R2 = R_weld @ urdf.rpy_to_rotation((0.1, 0.0, 0.0))
    p2 = R_weld @ np.array([0.0, 0.05, 0.15]) + np.array([0.2, 0.0, 0.1])
    assert np.abs(jt.E_tree() - R2.T).max() < 1e-12 and np.abs(np.array(jt.xyz) - p2).max() < 1e-12
    # the rotational inertia of the merged body is symmetric positive definite
    assert np.all(np.linalg.eigvalsh(I1) > 0)


def test_negative_axis_equals_mirrored_coordinates():
    """A joint about -z is the same mechanism as the joint about +z with q, qd, qdd and the torque negated."""
    plus = urdf.load_urdf(CHAIN % dict(sign=""))
    minus = urdf.load_urdf(CHAIN % dict(sign="-"))
    assert minus.get_S_ind_by_id(0) == 2
    q, qd, u = (a.astype(np.float64) for a in make_inputs(2, 5, 3))
    flip = np.array([-1.0, 1.0])
    tau_minus = _torques(minus, q, qd, u)
    tau_plus = _torques(plus, q * flip, qd * flip, u * flip)
    scale = np.abs(tau_plus).max()
    # damping acts on the velocity in both descriptions (same sign flip)
    assert np.abs(tau_minus - tau_plus * flip).max() < 1e-12 * max(1.0, scale)


def test_generator_accepts_a_loaded_robot(tmp_path, monkeypatch):
    from gridcodegenerator_amd import GRiDCodeGenerator
    robot = urdf.load_urdf(CHAIN % dict(sign="-"))
    monkeypatch.chdir(tmp_path)
    gen = GRiDCodeGenerator(robot, DEBUG_MODE=False, FILE_NAMESPACE="grid_t")
    gen.gen_all_code()
    assert (tmp_path / "grid_t.hip.h").exists() and "forward_dynamics_gradient_kernel" in gen.code_str


def test_unsupported_input():
    with pytest.raises(NotImplementedError):
        urdf.load_urdf(CHAIN.replace('<axis xyz="0 1 0"/>', '<axis xyz="0 0.6 0.8"/>') % dict(sign=""))
    with pytest.raises(NotImplementedError):
        urdf.load_urdf(CHAIN.replace('type="continuous"', 'type="floating"') % dict(sign=""))
    with pytest.raises(ValueError):
        urdf.load_urdf("<notrobot/>")


# ---------------------------------------------------------------------------------------------------------------------
# a URDF text written by hand in the style of a vendor description (NOT by this package's exporter), checked against an
# independent forward-kinematics / energy computation done right here from the XML
# ---------------------------------------------------------------------------------------------------------------------
def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def _rpy(r, p, y):
    return _rot((0, 0, 1), y) @ _rot((0, 1, 0), p) @ _rot((1, 0, 0), r)


class _PlainFK:
    """World poses of every link of a URDF for given joint angles -- straight from the XML, no spatial algebra."""

    def __init__(self, path):
        import xml.etree.ElementTree as ET
        root = ET.parse(path).getroot()
        num = lambda e, key, d: np.array([float(v) for v in (e.get(key) if e is not None and e.get(key) else d).split()])
        self.links = {}
        for l in root.findall("link"):
            i = l.find("inertial")
            if i is None:
                continue
            o, inr = i.find("origin"), i.find("inertia")
            I = np.array([[float(inr.get("ixx")), float(inr.get("ixy")), float(inr.get("ixz"))],
                          [float(inr.get("ixy")), float(inr.get("iyy")), float(inr.get("iyz"))],
                          [float(inr.get("ixz")), float(inr.get("iyz")), float(inr.get("izz"))]])
            self.links[l.get("name")] = (float(i.find("mass").get("value")), num(o, "xyz", "0 0 0"), _rpy(*num(o, "rpy", "0 0 0")), I)
        self.joints = []
        for j in root.findall("joint"):
            o, a = j.find("origin"), j.find("axis")
            self.joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                                    xyz=num(o, "xyz", "0 0 0"), R=_rpy(*num(o, "rpy", "0 0 0")), axis=num(a, "xyz", "1 0 0")))
        self.moving = [j["name"] for j in self.joints if j["type"] != "fixed"]

    def poses(self, q):
        q = dict(zip(self.moving, q))
        pose = {"world": (np.eye(3), np.zeros(3))}
        pending = list(self.joints)
        while pending:
            for j in list(pending):
                if j["parent"] in pose:
                    Rp, pp = pose[j["parent"]]
                    R = Rp @ j["R"]
                    if j["type"] != "fixed":
                        R = R @ _rot(j["axis"], q[j["name"]])
                    pose[j["child"]] = (R, Rp @ j["xyz"] + pp)
                    pending.remove(j)
        return pose

    def potential(self, q, g=9.81):
        pose = self.poses(q)
        return sum(m * g * (pose[name][0] @ c + pose[name][1])[2] for name, (m, c, Ri, I) in self.links.items() if name != "link_0")

    def kinetic(self, q, qd, eps=1e-6):
        plus, minus = self.poses(q + eps * qd), self.poses(q - eps * qd)
        now = self.poses(q)
        T = 0.0
        for name, (m, c, Ri, I) in self.links.items():
            v = ((plus[name][0] @ c + plus[name][1]) - (minus[name][0] @ c + minus[name][1])) / (2 * eps)
            W = (plus[name][0] - minus[name][0]) / (2 * eps) @ now[name][0].T
            w = np.array([W[2, 1], W[0, 2], W[1, 0]])
            Iw = now[name][0] @ Ri @ I @ Ri.T @ now[name][0].T
            T += 0.5 * m * v @ v + 0.5 * w @ Iw @ w
        return T


def test_hand_written_urdf_against_independent_energies():
    """tests/fixtures/lbr_iiwa14_like.urdf (world weld, alternating frames, a -z joint axis, rotated inertial frames, a
    flange + tool welded on): the loaded robot's gravity torques equal dU/dq and its mass matrix equals the Hessian of the
    kinetic energy, both computed from the XML by plain forward kinematics."""
    from urdf_fixture import FIXTURE
    robot = urdf.load_urdf(FIXTURE)
    fk = _PlainFK(FIXTURE)
    n = robot.get_num_joints()
    assert n == 7 and fk.moving == [robot.get_joint_by_id(j).get_name() for j in range(n)]
    T = O.RobotTables(robot)
    rng = np.random.default_rng(4)
    for _ in range(3):
        q = rng.uniform(-2.0, 2.0, n)
        tau_g = O.rnea(T, q[None], np.zeros((1, n)), np.zeros((1, n)))[0][0]
        dU = np.array([(fk.potential(q + 1e-6 * e) - fk.potential(q - 1e-6 * e)) / 2e-6 for e in np.eye(n)])
        assert np.abs(tau_g - dU).max() < 1e-6 * max(1.0, np.abs(dU).max())
        M = np.linalg.inv(O.minv(T, q[None], True)[0])
        Tk = lambda qd: fk.kinetic(q, qd)
        M_fd = np.array([[Tk(ei + ej) - Tk(ei) - Tk(ej) if i != j else 2 * Tk(ei) for j, ej in enumerate(np.eye(n))] for i, ei in enumerate(np.eye(n))])
        assert np.abs(M - M_fd).max() < 2e-6 * np.abs(M).max()
    assert [robot.get_damping_by_id(j) for j in range(n)] == [0.5, 0.5, 0.5, 0.5, 0.3, 0.3, 0.1]


@pytest.mark.gpu
def test_hand_written_urdf_on_gpu():
    """URDF file -> loader -> generator -> hipcc -> C ABI -> GPU, all five algorithms against the oracle on the same robot."""
    from urdf_fixture import register
    from gridcodegenerator_amd import host
    from test_gpu_parity import TOL, oracle_all, pack
    name = register()
    host.build_library(name, host.DEFAULT_PRECISION)
    robot = host.get_robot(name)
    T = O.RobotTables(robot)
    tol = TOL["iiwa7"]                                   # same size and mass range as the built-in 7-joint arm
    with host.GridHandle(name, device=0, precision=host.DEFAULT_PRECISION) as h:
        n, K = h.n, 333
        q, qd, u = make_inputs(n, K, 81)
        ref = oracle_all(T, q, qd, u)
        x = pack(q, qd, u)
        from conftest import relerr
        assert relerr(h.inverse_dynamics(x), ref["c"])[0] < tol["c"]
        assert relerr(h.direct_minv(x), ref["Minv"])[0] < tol["Minv"]
        assert relerr(h.forward_dynamics(x), ref["qdd"])[0] < tol["qdd"]
        assert relerr(h.inverse_dynamics_gradient(x), ref["dc_du_noqdd"])[0] < tol["dc_du"]
        assert relerr(h.forward_dynamics_gradient(x), ref["df_du"])[0] < tol["df_du"]
