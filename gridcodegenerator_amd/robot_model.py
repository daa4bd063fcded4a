"""Rigid-body robot model exposing the URDFParser-style robot API the generator consumes.

The reference generator is driven by a duck-typed ``robot`` object from the external
``URDFParser`` package (reference ``README.md:8,20``); the 28 members it touches are
listed in SURVEY.md section 8(b).  URDFParser is not available offline, so this module
is the build's own implementation of that contract:

* joints are numbered 0..n-1 in DFS pre-order (parent < child, subtree contiguous) --
  the reference silently relies on this (``algorithms/_direct_minv.py:141,150``,
  ``algorithms/_inverse_dynamics_gradient.py:505``);
* ``X_j(q) = X_J(q) * X_tree`` is the Featherstone parent->child *motion* transform
  ``[[E, 0], [-E r~, E]]`` (top-right block zero, bottom-right == top-left -- relied on by
  ``helpers/_topology_helpers.py:157,175-179``);
* the motion subspace ``S_j`` is a unit basis vector e_k (k in 0..2 revolute about x/y/z,
  3..5 prismatic along x/y/z) -- ``helpers/_topology_helpers.py:247,327``;
* spatial inertia is taken about the link-frame origin.

Nothing here is copied from the reference; it is the consumer-side contract restated.
"""
import math

import numpy as np

_AXIS_NAMES = "xyz"


def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rpy_to_rotation(rpy):
    """URDF fixed-axis roll/pitch/yaw -> frame rotation matrix R = Rz(y) Ry(p) Rx(r)."""
    r, p, y = rpy
    return rot_z(y) @ rot_y(p) @ rot_x(r)


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _snap(mat, tol=1e-12):
    """Snap values within tol of 0 / +-1 (axis-aligned rpy's give exact signed permutations)."""
    out = np.array(mat, dtype=np.float64)
    out[np.abs(out) < tol] = 0.0
    out[np.abs(out - 1.0) < tol] = 1.0
    out[np.abs(out + 1.0) < tol] = -1.0
    return out


def spatial_transform(E, r):
    """Motion transform [[E, 0], [-E r~, E]] (E: coordinate rotation, r: translation in parent frame)."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def spatial_inertia(mass, com, inertia_com):
    """6x6 spatial inertia about the link-frame origin: [[Ic + m c~ c~^T, m c~], [m c~^T, m 1]]."""
    c = skew(np.asarray(com, dtype=np.float64))
    I = np.zeros((6, 6))
    I[:3, :3] = np.asarray(inertia_com, dtype=np.float64) + mass * (c @ c.T)
    I[:3, 3:] = mass * c
    I[3:, :3] = mass * c.T
    I[3:, 3:] = mass * np.eye(3)
    return I


class Joint:
    """One single-DoF joint plus its child link (fixed joints are not represented)."""

    def __init__(self, name, parent, axis, jtype="revolute", xyz=(0.0, 0.0, 0.0), rpy=(0.0, 0.0, 0.0),
                 damping=0.0, link_name=None, mass=1.0, com=(0.0, 0.0, 0.0),
                 inertia=(1.0, 0.0, 0.0, 1.0, 0.0, 1.0)):
        if jtype not in ("revolute", "prismatic"):
            raise ValueError("joint type must be revolute or prismatic, got %r" % (jtype,))
        if axis not in (0, 1, 2):
            raise ValueError("joint axis must be 0, 1 or 2 (x, y, z); got %r" % (axis,))
        self.name = name
        self.parent = parent  # parent *joint name* (or None for a joint attached to the fixed base)
        self.axis = int(axis)
        self.jtype = jtype
        self.xyz = tuple(float(v) for v in xyz)
        self.rpy = tuple(float(v) for v in rpy)
        self.damping = float(damping)
        self.link_name = link_name if link_name is not None else name + "_link"
        self.mass = float(mass)
        self.com = tuple(float(v) for v in com)
        ixx, ixy, ixz, iyy, iyz, izz = (float(v) for v in inertia)
        self.inertia_com = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        # filled in by RobotModel
        self.jid = None
        self.parent_id = None

    # URDFParser-style accessors used for comments only (reference _inverse_dynamics.py:96-97)
    def get_name(self):
        return self.name

    def S_index(self):
        return self.axis + (3 if self.jtype == "prismatic" else 0)

    def E_tree(self):
        return _snap(rpy_to_rotation(self.rpy).T)

    def X_tree(self):
        return _snap(spatial_transform(self.E_tree(), np.asarray(self.xyz)))

    def X_joint(self, q):
        """X_J(q): coordinate rotation about / translation along the joint axis."""
        if self.jtype == "revolute":
            E = (rot_x, rot_y, rot_z)[self.axis](q).T
            return spatial_transform(E, np.zeros(3))
        r = np.zeros(3)
        r[self.axis] = q
        return spatial_transform(np.eye(3), r)

    def X(self, q):
        return self.X_joint(float(q)) @ self.X_tree()

    def X_basis(self):
        """Coefficient matrices (A, B, D, C) with X(q) = A sin q + B cos q + D q + C (exact)."""
        Xt = self.X_tree()
        A = np.zeros((6, 6)); B = np.zeros((6, 6)); D = np.zeros((6, 6)); C = np.zeros((6, 6))
        if self.jtype == "revolute":
            k = self.axis
            i, j = (k + 1) % 3, (k + 2) % 3
            # coordinate rotation about axis k: E[i,i]=c, E[i,j]=s, E[j,i]=-s, E[j,j]=c, E[k,k]=1
            Es = np.zeros((3, 3)); Ec = np.zeros((3, 3)); E1 = np.zeros((3, 3))
            Es[i, j] = 1.0; Es[j, i] = -1.0
            Ec[i, i] = 1.0; Ec[j, j] = 1.0
            E1[k, k] = 1.0
            z = np.zeros(3)
            A = spatial_transform(Es, z) @ Xt
            B = spatial_transform(Ec, z) @ Xt
            C = spatial_transform(E1, z) @ Xt
        else:
            e = np.zeros(3); e[self.axis] = 1.0
            Dm = np.zeros((6, 6)); Dm[3:, :3] = -skew(e)
            D = Dm @ Xt
            C = Xt.copy()
        return _snap(A), _snap(B), _snap(D), _snap(C)

    def spatial_inertia(self):
        return spatial_inertia(self.mass, self.com, self.inertia_com)


class Link:
    def __init__(self, name):
        self.name = name

    def get_name(self):
        return self.name


class RobotModel:
    """URDFParser-``Robot``-compatible model (see module docstring and SURVEY.md section 8(b))."""

    def __init__(self, name, joints, base_link_name="base_link"):
        self.name = name
        self._base_link = Link(base_link_name)
        self._joints = self._dfs_preorder(list(joints))
        self._n = len(self._joints)
        self._parents = [j.parent_id for j in self._joints]
        n = self._n
        self._children = [[c for c in range(n) if self._parents[c] == j] for j in range(n)]
        self._ancestors = []
        for j in range(n):
            anc = []
            p = self._parents[j]
            while p != -1:
                anc.append(p)
                p = self._parents[p]
            self._ancestors.append(sorted(anc))
        self._subtree = [[k for k in range(n) if k == j or j in self._ancestors[k]] for j in range(n)]
        for j in range(n):  # DFS pre-order invariant the whole code base relies on
            assert self._subtree[j] == list(range(j, j + len(self._subtree[j])))
        self._bfs_level = [len(self._ancestors[j]) for j in range(n)]
        self._Imats = [np.zeros((6, 6))] + [jt.spatial_inertia() for jt in self._joints]
        self._sympy_X = None

    @staticmethod
    def _dfs_preorder(joints):
        by_name = {}
        for j in joints:
            if j.name in by_name:
                raise ValueError("duplicate joint name " + j.name)
            by_name[j.name] = j
        kids = {None: []}
        for j in joints:
            kids.setdefault(j.name, [])
        for j in joints:
            if j.parent is not None and j.parent not in by_name:
                raise ValueError("joint %s has unknown parent %s" % (j.name, j.parent))
            kids[j.parent].append(j)
        order = []

        def visit(j, pid):
            j.jid = len(order)
            j.parent_id = pid
            order.append(j)
            for c in kids[j.name]:
                visit(c, j.jid)

        for root in kids[None]:
            visit(root, -1)
        if len(order) != len(joints):
            raise ValueError("joint graph is not a forest rooted at the base")
        return order

    # ---- sizes / topology ---------------------------------------------------------------
    def get_num_pos(self):
        return self._n

    def get_num_joints(self):
        return self._n

    def get_parent_id(self, jid):
        return self._parents[jid]

    def get_parent_id_array(self):
        return list(self._parents)

    def get_children_by_id(self, jid):
        return list(self._children[jid])

    def is_serial_chain(self):
        return all(self._parents[j] == j - 1 for j in range(self._n))

    def get_bfs_level_by_id(self, jid):
        return self._bfs_level[jid]

    def get_ids_by_bfs_level(self, level):
        return [j for j in range(self._n) if self._bfs_level[j] == level]

    def get_max_bfs_level(self):
        return max(self._bfs_level)

    def get_max_bfs_width(self):
        return max(len(self.get_ids_by_bfs_level(l)) for l in range(self.get_max_bfs_level() + 1))

    def get_ancestors_by_id(self, jid):
        return list(self._ancestors[jid])  # fresh list: callers mutate it (reference _test.py:355-356)

    def get_subtree_by_id(self, jid):
        return list(self._subtree[jid])  # includes jid, ascending, contiguous

    def get_total_ancestor_count(self):
        return sum(len(a) for a in self._ancestors)

    def get_total_subtree_count(self):
        return sum(len(s) for s in self._subtree)

    def get_is_ancestor_of(self, jid, jid_of_interest):
        """True if ``jid`` is an ancestor of ``jid_of_interest``."""
        return jid in self._ancestors[jid_of_interest]

    def get_is_in_subtree_of(self, jid, jid_of_interest):
        """True if ``jid`` lies in the subtree rooted at ``jid_of_interest``."""
        return jid in self._subtree[jid_of_interest]

    def get_unique_parent_ids(self, jids):
        return sorted(set(self._parents[j] for j in jids))

    def has_repeated_parents(self, jids):
        ps = [self._parents[j] for j in jids]
        return len(ps) != len(set(ps))

    # ---- joints / links -----------------------------------------------------------------
    def get_joint_by_id(self, jid):
        return self._joints[jid]

    def get_link_by_id(self, lid):
        return Link(self._joints[lid].link_name)

    def get_S_by_id(self, jid):
        S = np.zeros(6)
        S[self._joints[jid].S_index()] = 1
        return S

    def get_S_ind_by_id(self, jid):
        return self._joints[jid].S_index()

    def are_Ss_identical(self, jids):
        return len(set(self._joints[j].S_index() for j in jids)) <= 1

    def get_damping_by_id(self, jid):
        return self._joints[jid].damping

    # ---- transforms -----------------------------------------------------------------------
    def get_Xmat_Func_by_id(self, jid):
        return self._joints[jid].X

    def get_Xmat_Funcs_ordered_by_id(self):
        return [jt.X for jt in self._joints]

    def get_Xmat_basis_by_id(self, jid):
        return self._joints[jid].X_basis()

    def get_Xmats_ordered_by_id(self):
        """sympy 6x6 matrices in a symbol printed as ``theta`` (reference helpers/_topology_helpers.py:22-32)."""
        if self._sympy_X is None:
            import sympy as sp
            theta = sp.symbols("theta")

            def coef(x):
                return int(x) if abs(x) == 1.0 else sp.Float(x)

            mats = []
            for jt in self._joints:
                A, B, D, C = jt.X_basis()
                M = sp.zeros(6, 6)
                for r in range(6):
                    for c in range(6):
                        e = sp.Integer(0)
                        if A[r, c] != 0.0:
                            e = e + coef(A[r, c]) * sp.sin(theta)
                        if B[r, c] != 0.0:
                            e = e + coef(B[r, c]) * sp.cos(theta)
                        if D[r, c] != 0.0:
                            e = e + coef(D[r, c]) * theta
                        if C[r, c] != 0.0:
                            e = e + coef(C[r, c])
                        M[r, c] = e
                mats.append(M)
            self._sympy_X = mats
        return self._sympy_X

    # ---- inertias ---------------------------------------------------------------------------
    def get_Imats_ordered_by_id(self):
        return [m.copy() for m in self._Imats]  # n+1 entries, base first

    def get_Imat_by_id(self, jid):
        return self._Imats[jid + 1].copy()

    def get_Imats_dict_by_id(self):
        return {j: self._Imats[j + 1].copy() for j in range(self._n)}

    # ---- description --------------------------------------------------------------------------
    def describe(self):
        lines = ["robot %s: %d joints" % (self.name, self._n)]
        for jt in self._joints:
            lines.append("  %2d %-14s parent %2d  %s-%s" % (jt.jid, jt.name, jt.parent_id, jt.jtype,
                                                            _AXIS_NAMES[jt.axis]))
        return "\n".join(lines)
