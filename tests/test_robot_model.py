"""Integer topology / sparsity bookkeeping and model-constant tables: bit-exact against what the
reference computes for the same robot object (fixtures from tests/golden/make_golden.py)."""
import numpy as np

from gridcodegenerator_amd import GRiDCodeGenerator
from gridcodegenerator_amd.emit.model import RobotSpec


def test_sparsity_tables_bit_exact(robot_name, golden, robots):
    G = golden(robot_name)
    t = RobotSpec(robots(robot_name)).sparsity_tables()
    for key in ("dva_cols_per_jid", "running_sum_dva_cols_per_jid", "df_cols_per_jid", "running_sum_df_cols_per_jid",
                "df_col_that_is_jid", "num_ancestors", "num_subtree", "running_sum_num_ancestors", "running_sum_num_subtree"):
        assert list(t[key]) == list(G[key]), key
    assert t["dva_cols_per_partial"] == int(G["dva_cols_per_partial"])
    assert t["df_cols_per_partial"] == int(G["df_cols_per_partial"])


def test_generator_python_api_matches_reference(robot_name, golden, robots):
    G = golden(robot_name)
    gen = GRiDCodeGenerator(robots(robot_name))
    assert gen.gen_topology_helpers_size() == int(G["topology_helpers_size"])
    out = gen.gen_topology_sparsity_helpers_python()
    assert out[0] == int(G["dva_cols_per_partial"]) and out[3] == int(G["df_cols_per_partial"])
    assert list(out[1]) == list(G["dva_cols_per_jid"]) and list(out[6]) == list(G["df_col_that_is_jid"])
    init = gen.gen_topology_sparsity_helpers_python(True)
    assert [int(x) for x in init[0]] == list(G["num_ancestors"]) and all(isinstance(x, str) for x in init[0])


def test_topology_helpers_row_bit_exact(robot_name, golden, robots):
    G = golden(robot_name)
    row = RobotSpec(robots(robot_name)).topology_helpers_row()
    assert row == [int(x) for x in G["h_topology_helpers"]]
    assert len(row) == int(G["topology_helpers_size"])


def test_reference_size_constants_known_answers(robot_name, golden, robots):
    G = golden(robot_name)
    assert RobotSpec(robots(robot_name)).reference_size_constants() == [int(x) for x in G["size_constants"]]


def test_known_answers_from_survey(robots):
    assert RobotSpec(robots("iiwa7")).reference_size_constants() == [546, 1171, 1220, 2226, 2226, 2471, 2625, 352]
    assert RobotSpec(robots("atlas30")).reference_size_constants() == [2340, 9210, 10110, 10980, 10980, 13410, 16140, 512]


def test_XImats_table_matches_reference_emit(robot_name, golden, robots):
    G = golden(robot_name)
    table = RobotSpec(robots(robot_name)).XImats_table()
    assert table.shape == G["h_XImats"].shape
    assert np.abs(table - G["h_XImats"]).max() < 1e-12


def test_robot_api_contract(robot_name, robots):
    r = robots(robot_name)
    n = r.get_num_pos()
    anc = r.get_ancestors_by_id(n - 1)
    anc.append(999)  # callers mutate the returned list (reference _test.py:355-356)
    assert 999 not in r.get_ancestors_by_id(n - 1)
    for j in range(n):
        assert r.get_parent_id(j) < j
        sub = r.get_subtree_by_id(j)
        assert sub[0] == j and sub == list(range(j, j + len(sub)))
        assert np.asarray(r.get_S_by_id(j)).tolist().count(1) == 1
        X = r.get_Xmat_Func_by_id(j)(0.37)
        assert np.abs(X[:3, 3:]).max() == 0.0 and np.abs(X[:3, :3] - X[3:, 3:]).max() < 1e-15
    assert len(r.get_Imats_ordered_by_id()) == n + 1
    assert len(r.get_Xmats_ordered_by_id()) == n


def test_sympy_X_agrees_with_numeric_X(robots):
    import sympy as sp
    r = robots("mixed5")
    theta = sp.symbols("theta")
    for j, M in enumerate(r.get_Xmats_ordered_by_id()):
        f = sp.lambdify(theta, M, "numpy")
        assert np.abs(np.array(f(0.7), dtype=float) - r.get_Xmat_Func_by_id(j)(0.7)).max() < 1e-12
