"""Built-in robot models (no URDF files exist offline, so constants are embedded).

``iiwa7``   -- 7-DoF KUKA LBR iiwa-14-like serial chain, every joint revolute about z
              (``are_Ss_identical`` and ``is_serial_chain`` are both True: the reference then
              needs no topology table, helpers/_topology_helpers.py:217-226).
``quad12``  -- 12-DoF HyQ-like quadruped on a fixed trunk: four 3-joint legs, i.e. FOUR base-rooted trees (DFS pre-order
              ids LF 0-2, RF 3-5, LH 6-8, RH 9-11); sits on the n = 12 switch between the fused and the column-serial schedule.
``atlas30`` -- 30-DoF Atlas-v5-like branched humanoid on a fixed pelvis, DFS pre-order ids:
              back 0-2, l_arm 3-9, neck 10, r_arm 11-17, l_leg 18-23, r_leg 24-29.

Kinematic skeletons follow the public URDF conventions (origin xyz / rpy per joint); inertial
constants are plausible approximations -- parity only needs both sides of a comparison to
share one model object.
"""
import math

from .robot_model import Joint, RobotModel

_PI = math.pi


def iiwa7():
    h = _PI / 2
    spec = [
        # name, xyz, rpy, mass, com, inertia(ixx, ixy, ixz, iyy, iyz, izz)
        ("iiwa_joint_1", (0.0, 0.0, 0.1575), (0.0, 0.0, 0.0), 5.76, (0.0, -0.03, 0.12), (0.033, 0.0, 0.0, 0.0333, 0.0, 0.0123)),
        ("iiwa_joint_2", (0.0, 0.0, 0.2025), (h, 0.0, _PI), 6.35, (0.0003, 0.059, 0.042), (0.0305, 0.0, 0.0, 0.0304, 0.0, 0.011)),
        ("iiwa_joint_3", (0.0, 0.2045, 0.0), (h, 0.0, _PI), 3.5, (0.0, 0.03, 0.13), (0.025, 0.0, 0.0, 0.0238, 0.0, 0.0076)),
        ("iiwa_joint_4", (0.0, 0.0, 0.2155), (h, 0.0, 0.0), 3.5, (0.0, 0.067, 0.034), (0.017, 0.0, 0.0, 0.0164, 0.0, 0.006)),
        ("iiwa_joint_5", (0.0, 0.1845, 0.0), (-h, _PI, 0.0), 3.5, (0.0001, 0.021, 0.076), (0.01, 0.0, 0.0, 0.0087, 0.0, 0.00449)),
        ("iiwa_joint_6", (0.0, 0.0, 0.2155), (h, 0.0, 0.0), 1.8, (0.0, 0.0006, 0.0004), (0.0049, 0.0, 0.0, 0.0047, 0.0, 0.0036)),
        ("iiwa_joint_7", (0.0, 0.081, 0.0), (-h, _PI, 0.0), 1.2, (0.0, 0.0, 0.02), (0.001, 0.0, 0.0, 0.001, 0.0, 0.001)),
    ]
    joints = []
    parent = None
    for i, (name, xyz, rpy, mass, com, inertia) in enumerate(spec):
        joints.append(Joint(name, parent, axis=2, jtype="revolute", xyz=xyz, rpy=rpy, damping=0.0,
                            link_name="iiwa_link_%d" % (i + 1), mass=mass, com=com, inertia=inertia))
        parent = name
    return RobotModel("iiwa7", joints, base_link_name="iiwa_link_0")


def _atlas_arm(side, sgn):
    # (joint, axis, xyz, link, mass, com, inertia) ; sgn mirrors y for the right side
    p = side + "_arm_"
    return [
        (p + "shz", 2, (0.1406, sgn * 0.2256, 0.4776), side + "_clav", 4.466, (0.0, sgn * 0.048, 0.084), (0.011, 0.0, 0.0, 0.009, sgn * -0.004, 0.004)),
        (p + "shx", 0, (0.0, sgn * 0.11, 0.245), side + "_scap", 3.899, (0.0, 0.0, 0.0), (0.00319, 0.0, 0.0, 0.00583, 0.0, 0.00583)),
        (p + "ely", 1, (0.0, sgn * 0.187, 0.016), side + "_uarm", 4.386, (0.0, sgn * -0.065, 0.0), (0.00656, 0.0, 0.0, 0.00358, 0.0, 0.00656)),
        (p + "elx", 0, (0.0, sgn * 0.119, 0.0092), side + "_larm", 3.248, (0.0, 0.0, 0.0), (0.00265, 0.0, 0.0, 0.00446, 0.0, 0.00446)),
        (p + "wry", 1, (0.0, sgn * 0.29955, -0.00921), side + "_ufarm", 2.4798, (0.00015, sgn * 0.08296, 0.00037), (0.012731, 0.0, 0.0, 0.002857, 0.0, 0.011948)),
        (p + "wrx", 0, (0.0, 0.0, 0.0), side + "_lfarm", 0.648, (0.00017, sgn * -0.02515, 0.00163), (0.000764, 0.0, 0.0, 0.000429, 0.0, 0.000825)),
        (p + "wry2", 1, (0.0, sgn * 0.051, 0.0), side + "_hand", 0.5839, (0.0016, sgn * 0.07, 0.0001), (0.000388, 0.0, 0.0, 0.000477, 0.0, 0.000379)),
    ]


def _atlas_leg(side, sgn):
    p = side + "_leg_"
    return [
        (p + "hpz", 2, (0.0, sgn * 0.089, 0.0), side + "_uglut", 1.959, (0.00529, sgn * -0.00344, 0.00313), (0.00074276, 0.0, -2.79549e-05, 0.000688179, 0.0, 0.00041242)),
        (p + "hpx", 0, (0.0, 0.0, 0.0), side + "_lglut", 0.898, (0.0133, sgn * 0.017, -0.0312), (0.000691326, sgn * -2.24344e-05, 2.50508e-06, 0.00126856, sgn * 0.000137862, 0.00106487)),
        (p + "hpy", 1, (0.05, sgn * 0.0225, -0.066), side + "_uleg", 8.204, (0.0, 0.0, -0.21), (0.09, 0.0, 0.0, 0.09, 0.0, 0.02)),
        (p + "kny", 1, (-0.05, 0.0, -0.374), side + "_lleg", 4.515, (0.001, 0.0, -0.187), (0.077, 0.0, -0.003, 0.076, 0.0, 0.01)),
        (p + "aky", 1, (0.0, 0.0, -0.422), side + "_talus", 0.125, (0.0, 0.0, 0.0), (1.01674e-05, 0.0, 0.0, 8.42775e-06, 0.0, 1.30101e-05)),
        (p + "akx", 0, (0.0, 0.0, 0.0), side + "_foot", 2.41, (0.027, 0.0, -0.067), (0.002, 0.0, 0.0, 0.007, 0.0, 0.008)),
    ]


def atlas30():
    joints = []

    def chain(spec, first_parent):
        parent = first_parent
        for (name, axis, xyz, link, mass, com, inertia) in spec:
            joints.append(Joint(name, parent, axis=axis, jtype="revolute", xyz=xyz, rpy=(0.0, 0.0, 0.0),
                                damping=0.0, link_name=link, mass=mass, com=com, inertia=inertia))
            parent = name

    back = [
        ("back_bkz", 2, (-0.0125, 0.0, 0.0), "ltorso", 2.27, (-0.0112984, -3.15366e-06, 0.0746835), (0.0039092, -5.04491e-08, -0.000342157, 0.00341694, 4.87119e-07, 0.00174492)),
        ("back_bky", 1, (0.0, 0.0, 0.162), "mtorso", 0.799, (-0.00816266, -0.0131245, 0.0305974), (0.000454181, -6.10764e-05, 3.94009e-05, 0.000483282, 5.27463e-05, 0.000444215)),
        ("back_bkx", 0, (0.0, 0.0, 0.05), "utorso", 84.409, (-0.0622, 0.0023, 0.3157), (1.577, -0.032, 0.102, 1.602, 0.047, 0.565)),
    ]
    chain(back, None)
    chain(_atlas_arm("l", 1.0), "back_bkx")
    chain([("neck_ry", 1, (0.2546, 0.0, 0.6215), "head", 1.4199, (-0.075, 3.3e-05, 0.0277), (0.0039688, -1.5797e-06, -0.00089293, 0.0041178, -6.8415e-07, 0.0035243))], "back_bkx")
    chain(_atlas_arm("r", -1.0), "back_bkx")
    chain(_atlas_leg("l", 1.0), None)
    chain(_atlas_leg("r", -1.0), None)
    return RobotModel("atlas30", joints, base_link_name="pelvis")


def chain_prismatic_test_robot():
    """Small 4-joint mixed revolute/prismatic branched robot used only by tests (exercises S_ind 0..5)."""
    joints = [
        Joint("j0", None, axis=2, jtype="revolute", xyz=(0.0, 0.0, 0.1), rpy=(0.0, 0.0, 0.0), mass=2.0,
              com=(0.01, 0.02, 0.05), inertia=(0.02, 0.001, 0.0, 0.03, 0.0, 0.01)),
        Joint("j1", "j0", axis=0, jtype="prismatic", xyz=(0.1, 0.0, 0.2), rpy=(0.3, -0.2, 0.5), mass=1.5,
              com=(0.0, 0.03, 0.02), inertia=(0.01, 0.0, 0.0005, 0.012, 0.0, 0.008), damping=0.1),
        Joint("j2", "j1", axis=1, jtype="revolute", xyz=(0.0, 0.15, 0.0), rpy=(_PI / 2, 0.0, 0.0), mass=1.0,
              com=(0.02, 0.0, 0.04), inertia=(0.005, 0.0, 0.0, 0.006, 0.0002, 0.004)),
        Joint("j3", "j0", axis=1, jtype="prismatic", xyz=(-0.1, 0.05, 0.1), rpy=(0.0, 0.4, 0.0), mass=0.8,
              com=(0.0, 0.0, 0.03), inertia=(0.003, 0.0, 0.0, 0.003, 0.0, 0.002), damping=0.05),
        Joint("j4", None, axis=0, jtype="revolute", xyz=(0.0, -0.2, 0.0), rpy=(0.0, 0.0, 0.7), mass=1.2,
              com=(0.03, 0.0, 0.0), inertia=(0.004, 0.0, 0.0, 0.005, 0.0, 0.006)),
    ]
    return RobotModel("mixed5", joints)


def quad12():
    """HyQ-like quadruped on a fixed trunk: four legs of three revolute joints (hip abduction about x, hip flexion and knee about y),
    each leg a base-rooted tree of its own -- 12 joints, 4 trees, DFS pre-order ids LF 0-2, RF 3-5, LH 6-8, RH 9-11.  n = 12 is the
    largest robot on the fused gradient schedule (larger ones use the column-serial recomputing schedule), and four trees exercise the
    tree-wise machinery (block-diagonal Minv, wave-per-configuration groups) that a chain and a humanoid do not."""
    joints = []
    for leg, (sx, sy) in (("lf", (1.0, 1.0)), ("rf", (1.0, -1.0)), ("lh", (-1.0, 1.0)), ("rh", (-1.0, -1.0))):
        spec = [
            (leg + "_haa", 0, (sx * 0.3735, sy * 0.207, 0.0), (0.0, 0.0, 0.0), leg + "_hipassembly", 2.93, (sx * 0.04263, sy * 0.0, -0.16931),
             (0.05071, sx * sy * 4e-05, sx * 0.00159, 0.05486, sy * -5e-05, 0.00571)),
            (leg + "_hfe", 1, (sx * 0.08, 0.0, 0.0), (0.0, 0.0, 0.0), leg + "_upperleg", 2.638, (sx * 0.01, sy * 0.002, -0.15074),
             (0.08873, sx * sy * -0.00023, sx * 0.00067, 0.09176, sy * 0.00005, 0.00342)),
            (leg + "_kfe", 1, (0.0, 0.0, -0.35), (0.0, 0.0, 0.0), leg + "_lowerleg", 0.881, (sx * 0.005, 0.0, -0.1254),
             (0.02213, 0.0, sx * 0.0001, 0.02218, 0.0, 0.00012)),
        ]
        parent = None
        for (name, axis, xyz, rpy, link, mass, com, inertia) in spec:
            joints.append(Joint(name, parent, axis=axis, jtype="revolute", xyz=xyz, rpy=rpy, damping=0.0,
                                link_name=link, mass=mass, com=com, inertia=inertia))
            parent = name
    return RobotModel("quad12", joints, base_link_name="trunk")


BUILTIN_ROBOTS = {"iiwa7": iiwa7, "atlas30": atlas30, "mixed5": chain_prismatic_test_robot, "quad12": quad12}


REGISTERED_ROBOTS = {}


def register_robot(name, factory):
    """Make a robot object (e.g. one loaded from a URDF file with gridcodegenerator_amd.urdf.load_urdf) available to
    host.build_library / GridHandle under `name`: factory() -> robot object."""
    if name in BUILTIN_ROBOTS:
        raise ValueError("%r is a built-in robot" % name)
    if not name.replace("_", "").isalnum():
        raise ValueError("robot names become C++ namespaces and file names: letters, digits and underscores only")
    REGISTERED_ROBOTS[name] = factory


def register_urdf(name, path):
    from .urdf import load_urdf
    register_robot(name, lambda: load_urdf(path))


def get_robot(name):
    if name in BUILTIN_ROBOTS:
        return BUILTIN_ROBOTS[name]()
    if name in REGISTERED_ROBOTS:
        return REGISTERED_ROBOTS[name]()
    raise KeyError("unknown robot %r (built in: %s; registered: %s)" % (name, sorted(BUILTIN_ROBOTS), sorted(REGISTERED_ROBOTS)))
