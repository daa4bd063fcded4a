"""Generation-time view of a URDFParser-style robot object (plain Python/numpy data).

Consumes only the duck-typed robot API the reference consumes:
``get_num_pos, get_parent_id, get_S_by_id, get_ancestors_by_id, get_subtree_by_id,
get_Xmat_Func_by_id, get_Imat_by_id, get_damping_by_id, ...``.

The reference reads X_j(q) as sympy expressions and string-substitutes ``sin(theta)``/``cos(theta)``
(helpers/_topology_helpers.py:159-170).  Here X_j(q) is recovered from the *numeric* callable
(``get_Xmat_Func_by_id``) by an exact fit X = A sin q + B cos q + D q + C -- a single-DoF revolute or
prismatic joint always has this form -- so generation needs no sympy and takes milliseconds.
"""
import numpy as np


def _clean(M, tol=1e-13):
    M = np.array(M, dtype=np.float64)
    M[np.abs(M) < tol] = 0.0
    for target in (1.0, -1.0):
        M[np.abs(M - target) < tol] = target
    return M


def fit_X_basis(func):
    """Return (A, B, D, C), each 6x6, with X(q) = A sin q + B cos q + D q + C."""
    th = np.array([-2.3, -1.1, -0.4, 0.3, 0.9, 1.7, 2.6])
    Phi = np.stack([np.sin(th), np.cos(th), th, np.ones_like(th)], axis=1)
    Y = np.stack([np.asarray(func(float(t)), dtype=np.float64).reshape(36) for t in th])
    coef, *_ = np.linalg.lstsq(Phi, Y, rcond=None)
    coef = _clean(coef)
    for t in (0.123, -2.9, 3.05):
        fit = (np.array([np.sin(t), np.cos(t), t, 1.0]) @ coef).reshape(6, 6)
        if np.abs(fit - np.asarray(func(t), dtype=np.float64)).max() > 1e-10:
            raise ValueError("X(q) is not of the form A sin q + B cos q + D q + C; unsupported joint type")
    A, B, D, C = (coef[i].reshape(6, 6) for i in range(4))
    return A, B, D, C


class RobotSpec:
    def __init__(self, robot):
        self.robot = robot
        self.name = getattr(robot, "name", "robot")
        n = robot.get_num_pos()
        self.n = n
        self.parent = [int(robot.get_parent_id(j)) for j in range(n)]
        self.S_ind = [int(np.asarray(robot.get_S_by_id(j)).tolist().index(1)) for j in range(n)]
        self.ancestors = [sorted(int(a) for a in robot.get_ancestors_by_id(j)) for j in range(n)]
        self.subtree = [sorted(int(a) for a in robot.get_subtree_by_id(j)) for j in range(n)]
        self.children = [[c for c in range(n) if self.parent[c] == j] for j in range(n)]
        self.damping = [float(robot.get_damping_by_id(j)) for j in range(n)]
        self.Imats = [_clean(np.asarray(robot.get_Imat_by_id(j), dtype=np.float64), 1e-15) for j in range(n)]
        self.Xbasis = [fit_X_basis(robot.get_Xmat_Func_by_id(j)) for j in range(n)]
        self.joint_names = [self._name(robot.get_joint_by_id, j) for j in range(n)]
        self.link_names = [self._name(robot.get_link_by_id, j) for j in range(n)]
        for j in range(n):
            if self.parent[j] >= j:
                raise ValueError("joint ids must be DFS pre-order (parent < child)")
            if self.subtree[j] != list(range(j, j + len(self.subtree[j]))):
                raise ValueError("joint ids must be DFS pre-order (contiguous subtrees)")
            A, B, D, C = self.Xbasis[j]
            for M in (A, B, D, C):
                if np.abs(M[:3, 3:]).max() != 0.0 or np.abs(M[:3, :3] - M[3:, 3:]).max() > 1e-12:
                    raise ValueError("X must have the block form [[E,0],[-E r~,E]]")
        self.is_serial_chain = all(self.parent[j] == j - 1 for j in range(n))
        self.identical_S = len(set(self.S_ind)) <= 1
        self.uses_trig = [bool(np.abs(self.Xbasis[j][0]).max() > 0 or np.abs(self.Xbasis[j][1]).max() > 0)
                          for j in range(n)]
        self.uses_theta = [bool(np.abs(self.Xbasis[j][2]).max() > 0) for j in range(n)]

    @staticmethod
    def _name(getter, j):
        try:
            return str(getter(j).get_name())
        except Exception:
            return "id%d" % j

    # ---- integer bookkeeping of the reference (helpers/_topology_helpers.py:184-215) -------------
    def topology_helpers_size(self):
        size = 0
        if not self.is_serial_chain:
            size += 5 * self.n + 1
        if not self.identical_S:
            size += self.n
        return size

    def sparsity_tables(self):
        n = self.n
        num_anc = [len(self.ancestors[j]) for j in range(n)]
        num_sub = [len(self.subtree[j]) for j in range(n)]
        rs_anc = [sum(num_anc[:j]) for j in range(n + 1)]
        rs_sub = [sum(num_sub[:j]) for j in range(n)]
        return dict(
            num_ancestors=num_anc, num_subtree=num_sub,
            running_sum_num_ancestors=rs_anc, running_sum_num_subtree=rs_sub,
            dva_cols_per_partial=sum(num_anc) + n,
            df_cols_per_partial=sum(num_anc) + sum(num_sub),
            dva_cols_per_jid=[num_anc[j] + 1 for j in range(n)],
            df_cols_per_jid=[num_anc[j] + num_sub[j] for j in range(n)],
            df_col_that_is_jid=list(num_anc),
            running_sum_dva_cols_per_jid=[rs_anc[j] + j for j in range(n + 1)],
            running_sum_df_cols_per_jid=[rs_anc[j] + rs_sub[j] for j in range(n)],
        )

    def topology_helpers_row(self):
        """The int table the reference uploads as d_topology_helpers (helpers/_topology_helpers.py:236-251)."""
        t = self.sparsity_tables()
        row = []
        if not self.is_serial_chain:
            row += list(self.parent)
            if not self.identical_S:
                row += list(self.S_ind)
            row += t["num_ancestors"] + t["num_subtree"] + t["running_sum_num_ancestors"] + t["running_sum_num_subtree"]
        elif not self.identical_S:
            row += list(self.S_ind)
        return row

    def max_bfs_width(self):
        depth = [len(a) for a in self.ancestors]
        return max(depth.count(d) for d in set(depth))

    def reference_size_constants(self):
        """The shared-memory element counts the *reference* would emit (GRiDCodeGenerator.py:70-83).

        Kept as documentation / golden-checked bookkeeping; the lane-per-configuration kernels size
        their LDS differently (see GRiDCodeGenerator.gen_add_constants_helpers)."""
        n = self.n
        t = self.sparsity_tables()
        XI = 72 * n
        id_t = 6 * n
        minv_t = 6 * n * n + 36 * n + 6 * n + n + 72 * self.max_bfs_width()
        fd_t = minv_t + n * n
        iddu_t = 66 * n + 6 * (4 * t["dva_cols_per_partial"] + 2 * t["df_cols_per_partial"])
        fddu_t = max(minv_t, iddu_t)
        iddu_max = 2 * n + 2 * n * n + 18 * n + n + iddu_t
        fddu_max = 2 * n + 2 * n * n + 2 * n * n + 18 * n + n + n * n + n + fddu_t
        sugg = min(32 * int(np.ceil(12 * t["dva_cols_per_partial"] / 32.0)), 512)
        return [id_t + XI, minv_t + XI, fd_t + XI, iddu_t + XI, fddu_t + XI, iddu_max + XI, fddu_max + XI, sugg]

    def XImats_table(self):
        """72n values laid out as the reference's h_XImats (helpers/_topology_helpers.py:21-47):
        X_j constant parts (theta-dependent entries zero), column-major, then I_j column-major."""
        n = self.n
        out = np.zeros(72 * n)
        for j in range(n):
            A, B, D, C = self.Xbasis[j]
            dep = (A != 0) | (B != 0) | (D != 0)
            Xc = np.where(dep, 0.0, C)
            out[36 * j:36 * (j + 1)] = Xc.T.reshape(36)
            out[36 * (n + j):36 * (n + j + 1)] = self.Imats[j].T.reshape(36)
        return out


class SubForest:
    """The joints first..first+count-1 of a robot as a robot of their own (local ids 0..count-1): a run of consecutive base-rooted
    trees.  Base-rooted trees do not interact -- the joint-space inertia matrix, the bias forces and their gradients are block
    diagonal over them -- so a group of trees can be evaluated on its own (emit/wave.py: one wavefront per group)."""

    def __init__(self, spec, first, count):
        ids = list(range(first, first + count))
        for j in ids:
            assert spec.parent[j] == -1 or first <= spec.parent[j] < first + count, "a sub-forest must be closed under parents"
            assert all(first <= k < first + count for k in spec.subtree[j]), "a sub-forest must be closed under subtrees"
        self.full = spec
        self.first, self.n = first, count
        self.name = "%s[%d:%d]" % (spec.name, first, first + count)
        loc = lambda j: j - first
        self.parent = [(-1 if spec.parent[j] == -1 else loc(spec.parent[j])) for j in ids]
        self.S_ind = [spec.S_ind[j] for j in ids]
        self.ancestors = [[loc(a) for a in spec.ancestors[j]] for j in ids]
        self.subtree = [[loc(a) for a in spec.subtree[j]] for j in ids]
        self.children = [[loc(c) for c in spec.children[j]] for j in ids]
        self.damping = [spec.damping[j] for j in ids]
        self.Imats = [spec.Imats[j] for j in ids]
        self.Xbasis = [spec.Xbasis[j] for j in ids]
        self.uses_trig = [spec.uses_trig[j] for j in ids]
        self.uses_theta = [spec.uses_theta[j] for j in ids]


def base_trees(spec):
    """[(first joint, joint count)] of the base-rooted trees, in id order (DFS pre-order ids: each tree is a contiguous range)."""
    return [(j, len(spec.subtree[j])) for j in range(spec.n) if spec.parent[j] == -1]
