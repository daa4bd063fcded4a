"""The three build variants that failed on the GPU in round 1 (a memory fault, silently wrong numbers, wrong
results / hangs), rebuilt with the branch-free helpers of section 9.1 and checked against the oracle.  They are heavy spillers by
construction (hundreds of SGPR spills, kilobytes of scratch); what made them fail was spill code inside reduced-EXEC regions, and
the generated kernels have no such regions any more.  The variants are prebuilt by tools/build_regression_variants.py; a test skips
when its library is absent (they take up to half an hour each to compile) -- or FAILS when GRID_REQUIRE_REGRESSION_LIBS=1 (set by
the GPU run scripts, tools/r03_final.sh)."""
import os

import numpy as np
import pytest

from conftest import make_inputs, relerr
import regression_variants


def _lib(name):
    from gridcodegenerator_amd import host
    precision = regression_variants.register()[name]
    p = host.library_paths(name, precision)
    if not (os.path.exists(p["lib"]) and os.path.exists(p["stamp"])):
        msg = "variant %s not built (tools/build_regression_variants.py)" % name
        if os.environ.get("GRID_REQUIRE_REGRESSION_LIBS") == "1":      # set by the GPU run scripts: a clean checkout must not lose
            pytest.fail(msg)                                          # this coverage silently
        pytest.skip(msg)
    return precision, p


@pytest.mark.parametrize("name", sorted(regression_variants.VARIANTS))
def test_variant_is_what_failed_in_round_1_and_is_branch_free(name):
    """CPU: the variant really is a heavy spiller (the regime of the round-1 failures), and no kernel of it writes EXEC."""
    from gridcodegenerator_amd import host, isa_audit
    precision, p = _lib(name)
    res = host.kernel_resources(name, precision)
    assert max(k["sgpr_spills"] for k in res) >= 100 or max(k["scratch"] for k in res) >= 1024, "not the failing regime"
    audit = isa_audit.audit(p["lib"])
    assert {isa_audit.short_name(k): e for k, (n, e) in audit.items() if e} == {}


def _check_all(h, T, n, K, seed, tol, split_coop=True):
    from oracle import rbd_oracle as O
    q, qd, u = make_inputs(n, K, seed)
    q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
    x = np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))
    df, parts = O.fd_grad(T, q64, qd64, u64, return_parts=True)
    gflat = lambda M: np.concatenate([O.flat_colmajor(M[:, :, :n]), O.flat_colmajor(M[:, :, n:])], axis=1)
    qdd32 = np.ascontiguousarray(parts["qdd"].astype(np.float32))
    Minv_ref = O.flat_colmajor(np.triu(parts["Minv"]))
    errs = {
        "c": relerr(h.inverse_dynamics(x), parts["c"])[0],
        "Minv": relerr(h.direct_minv(x), Minv_ref)[0],
        "qdd": relerr(h.forward_dynamics(x), parts["qdd"])[0],
        "dc_du": relerr(h.inverse_dynamics_gradient(x), gflat(O.rnea_grad(T, q64, qd64, None)))[0],
        "dc_du_qdd": relerr(h.inverse_dynamics_gradient(x, qdd=qdd32), gflat(O.rnea_grad(T, q64, qd64, qdd32.astype(np.float64))))[0],
        "df_du": relerr(h.forward_dynamics_gradient(x), gflat(df))[0],
        "df_du_qdd_minv": relerr(h.forward_dynamics_gradient(x, qdd=qdd32, Minv=np.ascontiguousarray(Minv_ref.astype(np.float32))), gflat(df))[0],
    }
    for k, e in errs.items():
        assert e < tol.get(k, tol["dc_du"]), (k, e)
    return errs


@pytest.mark.gpu
def test_round1_a_register_capped_atlas_column_groups():
    """(a) `Memory access fault ... 0xffffffff5000`, Atlas-30, K = 16384, register-capped column groups (waves_per_simd = 2)."""
    import torch
    from gridcodegenerator_amd import host
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    from test_gpu_parity import TOL_BY_PRECISION
    precision, p = _lib("atlas30_capped")
    assert torch.cuda.is_available()
    T = O.RobotTables(get_robot("atlas30"))
    with host.GridHandle("atlas30_capped", precision=precision) as h:
        n = h.n
        errs = _check_all(h, T, n, 150, 5, TOL_BY_PRECISION["fp32"]["atlas30"])
        # the failing case itself: every column-split kernel of both gradients at K = 16384, bitwise against the unsplit kernel
        K = 16384
        q, qd, u = make_inputs(n, K, 9)
        d_in = torch.from_numpy(np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))).cuda()
        for alg, call in ((host.ALG_ID_DU, h.inverse_dynamics_gradient_device), (host.ALG_FD_DU, h.forward_dynamics_gradient_device)):
            h.set_coop(alg, 1) if alg == host.ALG_FD_DU else None
            outs = {}
            for S in [1] + h.L.splits(alg):
                h.set_split(alg, S)
                out = torch.full((K, 2 * n * n), float("nan"), dtype=torch.float32, device="cuda")
                call(out.data_ptr(), d_in.data_ptr(), 3 * n, K)
                h.synchronize()
                outs[S] = out.cpu().numpy()
                assert np.isfinite(outs[S]).all(), (alg, S)
            for S in outs:
                assert np.array_equal(outs[S], outs[1]) or relerr(outs[S], outs[1].astype(np.float64))[0] < 2e-6, (alg, S)
    print("round-1 (a) register-capped Atlas-30: %s" % {k: "%.1e" % v for k, v in errs.items()})


@pytest.mark.gpu
def test_round1_b_fused_schedule_30_joints():
    """(b) fused n > 12 kernels (2.5-4.7 KB of scratch in round 1): the USE_QDD_MINV forward-dynamics-gradient kernel returned
    wrong numbers (error 1.9 in dq columns 1-3)."""
    import torch
    from gridcodegenerator_amd import host
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    from test_gpu_parity import TOL_BY_PRECISION
    precision, p = _lib("atlas30_fused")
    assert torch.cuda.is_available()
    T = O.RobotTables(get_robot("atlas30"))
    with host.GridHandle("atlas30_fused", precision=precision) as h:
        errs = _check_all(h, T, h.n, 150, 5, TOL_BY_PRECISION["fp32"]["atlas30"])
        errs2 = _check_all(h, T, h.n, 2048, 6, TOL_BY_PRECISION["fp32"]["atlas30"])
    print("round-1 (b) fused Atlas-30: %s" % {k: "%.1e" % max(v, errs2[k]) for k, v in errs.items()})


@pytest.mark.gpu
def test_round1_c_all_double_arithmetic():
    """(c) fp64 kernels: wrong results / hangs in round 1.  All-double arithmetic with float I/O: every output to float rounding."""
    import torch
    from gridcodegenerator_amd import host
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    precision, p = _lib("iiwa7_fp64")
    assert torch.cuda.is_available()
    T = O.RobotTables(get_robot("iiwa7"))
    tol = dict(c=1.5e-7, Minv=1.5e-7, qdd=1.5e-7, dc_du=1.5e-7, dc_du_qdd=1.5e-7, df_du=1.5e-7, df_du_qdd_minv=6e-7)
    with host.GridHandle("iiwa7_fp64", precision=precision) as h:
        assert h.L.compute_dtype == "f64"
        errs = _check_all(h, T, h.n, 333, 5, tol)
        errs2 = _check_all(h, T, h.n, 4096, 6, tol)
    print("round-1 (c) all-double iiwa-7: %s" % {k: "%.1e" % max(v, errs2[k]) for k, v in errs.items()})


@pytest.mark.gpu
def test_all_double_atlas():
    """All-double arithmetic for the 30-joint robot (tests/regression_variants.py: EXTRA; a library nobody could build while
    heavy spillers miscomputed): every output to float rounding, i.e. the forward-dynamics gradient of Atlas-30 within
    north_star's 1e-6 on every batch -- at the price of a kernel that lives in scratch."""
    import torch
    from gridcodegenerator_amd import host
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    precision, p = _lib("atlas30_fp64")
    assert torch.cuda.is_available()
    T = O.RobotTables(get_robot("atlas30"))
    tol = dict(c=2e-7, Minv=2e-7, qdd=2e-7, dc_du=2e-7, dc_du_qdd=2e-7, df_du=3e-7, df_du_qdd_minv=1.7e-6)
    with host.GridHandle("atlas30_fp64", precision=precision) as h:
        assert h.L.compute_dtype == "f64"
        errs = _check_all(h, T, h.n, 333, 47, tol)      # (seed 47: the batch on which the mixed arithmetic reaches 1.1e-6)
        errs2 = _check_all(h, T, h.n, 2048, 6, tol)
    print("all-double Atlas-30: %s" % {k: "%.1e" % max(v, errs2[k]) for k, v in errs.items()})
