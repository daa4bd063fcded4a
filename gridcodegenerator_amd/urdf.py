"""Minimal URDF -> robot-object loader (and exporter) for the generator.

The reference is driven by a ``robot`` object produced by the external ``URDFParser`` package
(reference ``README.md:8,20``: ``URDFParser().parse("robot.urdf")``); that package is not
available offline, so this module provides the step *before* the hot path (SURVEY.md section 8(f),
rank 3) on top of ``robot_model.RobotModel``:

* ``load_urdf(source)`` -- ``source`` is a path or the XML text.  Supported: ``revolute`` /
  ``continuous`` / ``prismatic`` joints with an axis-aligned ``<axis>`` (a negative axis is absorbed
  into the child frame by a half turn about a perpendicular axis), ``fixed`` joints (the child link's
  inertia is merged into the moving body it is welded to, grandchildren are re-attached with the
  composed transform), ``<dynamics damping>``, inertial frames with a rotated ``<origin rpy>``.
  Joints are numbered in DFS pre-order following the order of the ``<joint>`` elements
  (parent < child, subtrees contiguous -- what the generator relies on).
  Not supported (``NotImplementedError``): ``floating`` / ``planar`` joints, skew joint axes.
* ``robot_to_urdf(robot)`` -- writes a ``RobotModel`` back as URDF text (used by the tests for
  round trips and to hand the built-in robots to other tools).

``URDFParser`` is a thin name-compatible front end: ``URDFParser().parse(path)``.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from .robot_model import Joint, RobotModel, rpy_to_rotation, spatial_inertia, spatial_transform

_AXIS_TOL = 1e-9


def _floats(text, count, default):
    if text is None:
        return np.array(default, dtype=np.float64)
    vals = [float(v) for v in text.split()]
    if len(vals) != count:
        raise ValueError("expected %d numbers, got %r" % (count, text))
    return np.array(vals, dtype=np.float64)


def _origin(elem):
    """<origin xyz rpy> of `elem` -> (R, p): pose of the described frame in its reference frame."""
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.eye(3), np.zeros(3)
    return rpy_to_rotation(_floats(o.get("rpy"), 3, (0.0, 0.0, 0.0))), _floats(o.get("xyz"), 3, (0.0, 0.0, 0.0))


def _compose(A, B):
    """Pose of frame C in frame A given B-in-A (A) and C-in-B (B)."""
    return A[0] @ B[0], A[0] @ B[1] + A[1]


def rotation_to_rpy(R):
    """Inverse of rpy_to_rotation (R = Rz(yaw) Ry(pitch) Rx(roll))."""
    sp = -R[2, 0]
    if abs(sp) > 1.0 - 1e-12:          # gimbal lock: pitch = +-pi/2, put everything into yaw
        pitch = math.copysign(math.pi / 2.0, sp)
        roll = 0.0
        yaw = math.atan2(-R[0, 1], R[1, 1])
    else:
        pitch = math.asin(sp)
        roll = math.atan2(R[2, 1], R[2, 2])
        yaw = math.atan2(R[1, 0], R[0, 0])
    return roll, pitch, yaw


def _link_inertia(link):
    """6x6 spatial inertia of a <link> about its link-frame origin (zeros without <inertial>)."""
    inertial = link.find("inertial")
    if inertial is None:
        return np.zeros((6, 6))
    R, c = _origin(inertial)
    mass_el, inertia_el = inertial.find("mass"), inertial.find("inertia")
    mass = float(mass_el.get("value")) if mass_el is not None else 0.0
    Ic = np.zeros((3, 3))
    if inertia_el is not None:
        g = lambda k: float(inertia_el.get(k, 0.0))
        Ic = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
    return spatial_inertia(mass, c, R @ Ic @ R.T)


def _move_inertia(I6, pose):
    """Spatial inertia given in frame B, expressed in frame A, with `pose` = (R, p) of B in A."""
    R, p = pose
    X = spatial_transform(R.T, p)       # motion transform A -> B coordinates
    return X.T @ I6 @ X


def _split_inertia(I6):
    """6x6 spatial inertia about the frame origin -> (mass, com, inertia about the com as ixx, ixy, ixz, iyy, iyz, izz)."""
    m = float(I6[5, 5])
    if m <= 0.0:
        return 0.0, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    mc = I6[:3, 3:]                     # m * skew(c)
    c = np.array([mc[2, 1], mc[0, 2], mc[1, 0]]) / m
    cx = np.array([[0.0, -c[2], c[1]], [c[2], 0.0, -c[0]], [-c[1], c[0], 0.0]])
    Ic = I6[:3, :3] - m * (cx @ cx.T)
    return m, tuple(c), (Ic[0, 0], Ic[0, 1], Ic[0, 2], Ic[1, 1], Ic[1, 2], Ic[2, 2])


def _axis_index(vec):
    a = np.asarray(vec, dtype=np.float64)
    nrm = np.linalg.norm(a)
    if nrm == 0.0:
        raise ValueError("zero joint axis")
    a = a / nrm
    k = int(np.argmax(np.abs(a)))
    e = np.zeros(3); e[k] = math.copysign(1.0, a[k])
    if np.abs(a - e).max() > _AXIS_TOL:
        raise NotImplementedError("joint axis %r is not aligned with x, y or z (the generated kernels use unit motion subspaces)" % (tuple(vec),))
    return k, a[k] < 0.0


def _half_turn(k):
    """Rotation by pi about the axis after k: maps e_k to -e_k (its own inverse)."""
    p = (k + 1) % 3
    F = -np.eye(3)
    F[p, p] = 1.0
    return F


def load_urdf(source, name=None):
    """Parse URDF (path or XML text) into a RobotModel.  See the module docstring for what is supported."""
    text = source
    if "<" not in source:
        with open(os.path.expanduser(source)) as fh:
            text = fh.read()
    root = ET.fromstring(text)
    if root.tag != "robot":
        raise ValueError("not a URDF document: root element is <%s>" % root.tag)
    links = {l.get("name"): l for l in root.findall("link")}
    joints = root.findall("joint")
    child_of = {}
    by_parent = {}
    for j in joints:
        parent, child = j.find("parent").get("link"), j.find("child").get("link")
        if parent not in links or child not in links:
            raise ValueError("joint %s references an unknown link" % j.get("name"))
        if child in child_of:
            raise ValueError("link %s has two parent joints (not a tree)" % child)
        child_of[child] = j
        by_parent.setdefault(parent, []).append(j)
    roots = [l for l in links if l not in child_of]
    if len(roots) != 1:
        raise ValueError("expected exactly one root link, found %r" % (roots,))

    bodies = []          # moving bodies in discovery (DFS) order: dict(name, parent, axis, jtype, pose, damping, link, I6)

    def walk(link_name, body, pose_in_body):
        """Attach link `link_name` (pose in the frame of moving body `body`, None = the fixed base) and descend."""
        if body is not None:
            body["I6"] += _move_inertia(_link_inertia(links[link_name]), pose_in_body)
        for j in by_parent.get(link_name, []):
            jtype = j.get("type")
            pose = _compose(pose_in_body, _origin(j))           # child (joint) frame in the body frame
            child = j.find("child").get("link")
            if jtype == "fixed":
                walk(child, body, pose)
                continue
            if jtype not in ("revolute", "continuous", "prismatic"):
                raise NotImplementedError("joint %s: type %r is not supported" % (j.get("name"), jtype))
            ax = j.find("axis")
            k, negative = _axis_index(_floats(ax.get("xyz") if ax is not None else None, 3, (1.0, 0.0, 0.0)))
            link_pose = (np.eye(3), np.zeros(3))                # child link frame in the new body's frame
            if negative:                                        # body frame = joint frame turned so that the axis is +e_k
                F = _half_turn(k)
                pose = (pose[0] @ F, pose[1])
                link_pose = (F.T, np.zeros(3))
            dyn = j.find("dynamics")
            new = dict(name=j.get("name"), parent=None if body is None else body["name"], axis=k,
                       jtype="prismatic" if jtype == "prismatic" else "revolute", pose=pose,
                       damping=float(dyn.get("damping", 0.0)) if dyn is not None else 0.0, link=child, I6=np.zeros((6, 6)))
            bodies.append(new)
            walk(child, new, link_pose)

    walk(roots[0], None, (np.eye(3), np.zeros(3)))
    if not bodies:
        raise ValueError("the URDF has no moving joints")
    out = []
    for b in bodies:
        mass, com, inertia = _split_inertia(b["I6"])
        out.append(Joint(b["name"], b["parent"], b["axis"], jtype=b["jtype"], xyz=tuple(b["pose"][1]),
                         rpy=rotation_to_rpy(b["pose"][0]), damping=b["damping"], link_name=b["link"],
                         mass=mass, com=com, inertia=inertia))
    return RobotModel(name or root.get("name") or "robot", out, base_link_name=roots[0])


def robot_to_urdf(robot, name=None):
    """URDF text of a RobotModel (one link per joint + the base link; inertial frames unrotated)."""
    n = robot.get_num_joints()
    fmt = lambda vals: " ".join(repr(float(v)) for v in vals)
    base = robot._base_link.get_name()
    lines = ['<?xml version="1.0"?>', '<robot name="%s">' % (name or robot.name), '  <link name="%s"/>' % base]
    for jid in range(n):
        jt = robot.get_joint_by_id(jid)
        Ic = jt.inertia_com
        lines += ['  <link name="%s">' % jt.link_name,
                  '    <inertial>',
                  '      <origin xyz="%s" rpy="0 0 0"/>' % fmt(jt.com),
                  '      <mass value="%r"/>' % float(jt.mass),
                  '      <inertia ixx="%r" ixy="%r" ixz="%r" iyy="%r" iyz="%r" izz="%r"/>'
                  % tuple(float(v) for v in (Ic[0, 0], Ic[0, 1], Ic[0, 2], Ic[1, 1], Ic[1, 2], Ic[2, 2])),
                  '    </inertial>',
                  '  </link>']
    for jid in range(n):
        jt = robot.get_joint_by_id(jid)
        parent_link = base if jt.parent_id == -1 else robot.get_joint_by_id(jt.parent_id).link_name
        axis = [0.0, 0.0, 0.0]; axis[jt.axis] = 1.0
        lines += ['  <joint name="%s" type="%s">' % (jt.name, jt.jtype),
                  '    <parent link="%s"/>' % parent_link,
                  '    <child link="%s"/>' % jt.link_name,
                  '    <origin xyz="%s" rpy="%s"/>' % (fmt(jt.xyz), fmt(jt.rpy)),
                  '    <axis xyz="%s"/>' % fmt(axis),
                  '    <dynamics damping="%r"/>' % float(jt.damping),
                  '    <limit lower="-3.14159" upper="3.14159" effort="1000" velocity="100"/>',
                  '  </joint>']
    lines.append('</robot>')
    return "\n".join(lines) + "\n"


class URDFParser:
    """Name-compatible front end: ``robot = URDFParser().parse("iiwa14.urdf")`` (reference README usage)."""

    def parse(self, filename, alpha_tie_breaker=False):
        if alpha_tie_breaker:
            raise NotImplementedError("children are visited in file order")
        return load_urdf(filename)
