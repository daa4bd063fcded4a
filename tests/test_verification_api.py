"""The reference-named CPU evaluation methods of the generator (test_rnea, test_minv, test_rnea_grad, test_fd_grad: IR
interpreter over the traced cores) against the golden fixtures produced by the reference's own methods."""
import os

import numpy as np
import pytest

from gridcodegenerator_amd import GRiDCodeGenerator
from gridcodegenerator_amd.robots import get_robot

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gens():
    return {name: GRiDCodeGenerator(get_robot(name)) for name in ("iiwa7", "atlas30")}


@pytest.mark.parametrize("name", ["iiwa7", "atlas30"])
def test_reference_named_methods_match_reference_outputs(name, gens):
    g = gens[name]
    d = np.load(os.path.join(GOLD, name + ".npz"))
    n = d["q"].shape[1]
    for k in range(2 if name == "atlas30" else 4):
        q, qd, u, qdd = d["q"][k], d["qd"][k], d["u"][k], d["qdd"][k]
        c0, _, _, _ = g.test_rnea(q, qd, None)
        assert np.abs(c0 - d["c_noqdd"][k]).max() < 1e-10 * max(1.0, np.abs(d["c_noqdd"][k]).max())
        c1, v, a, f = g.test_rnea(q, qd, qdd)
        assert np.abs(c1 - d["c_qdd"][k]).max() < 1e-10 * max(1.0, np.abs(d["c_qdd"][k]).max())
        for got, key in ((v, "v"), (a, "a"), (f, "f")):
            assert got.shape == (6, n)
            assert np.abs(got.T - d[key][k]).max() < 1e-10 * max(1.0, np.abs(d[key][k]).max())
        assert np.abs(g.test_minv(q, False) - d["Minv_upper"][k]).max() < 1e-9 * np.abs(d["Minv_dense"][k]).max()
        assert np.abs(g.test_minv(q) - d["Minv_dense"][k]).max() < 1e-9 * np.abs(d["Minv_dense"][k]).max()
        assert np.abs(g.test_rnea_grad(q, qd, None) - d["dc_du_noqdd"][k]).max() < 1e-9 * np.abs(d["dc_du_noqdd"][k]).max()
        assert np.abs(g.test_rnea_grad(q, qd, qdd) - d["dc_du_qdd"][k]).max() < 1e-9 * np.abs(d["dc_du_qdd"][k]).max()
        assert np.abs(g.test_fd_grad(q, qd, u) - d["df_du"][k]).max() < 1e-8 * np.abs(d["df_du"][k]).max()
        assert np.abs(g.test_forward_dynamics(q, qd, u) - qdd).max() < 1e-8 * np.abs(qdd).max()


def test_cross_product_helpers(gens):
    g = gens["iiwa7"]
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=6), rng.normal(size=6)
    w, v = a[:3], a[3:]
    # crm(a) b = [w x bw ; v x bw + w x bv],  crf(a) b = [w x bw + v x bv ; w x bv]
    assert np.allclose(g.mxv(a, b), np.concatenate([np.cross(w, b[:3]), np.cross(v, b[:3]) + np.cross(w, b[3:])]))
    assert np.allclose(g.fxv(a, b), np.concatenate([np.cross(w, b[:3]) + np.cross(v, b[3:]), np.cross(w, b[3:])]))
    e2 = np.zeros(6); e2[2] = 1.0
    assert np.allclose(g.mx2(a, 0.7), g.mxS(e2, a, 0.7)) and np.allclose(g.fxS(e2, a, 0.7), -g.mxS(e2, a, 0.7))
    assert np.allclose(g.mx(a), -g.fx(a).T)
