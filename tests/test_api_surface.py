"""The generated header's public surface that the C ABI does not reach, EXECUTED on the GPU (tests/api_surface_harness.hip):
every *_compute_only and *_launch wrapper, every USE_COMPRESSED_MEM=true instantiation, inverse_dynamics_vaf_device, the
_device tier inside a user kernel and the pointer-style _inner chain (load_update_XImats_helpers -> direct_minv_inner ->
inverse_dynamics_inner -> forward_dynamics_finish -> inverse_dynamics_inner_vaf -> inverse_dynamics_gradient_inner) run per lane
by a user kernel -- against the oracle.  Reference surface: GRiDCodeGenerator.py:243-279, algorithms/_inverse_dynamics.py:311-352,
423-495, _forward_dynamics_gradient.py:59-99.

Tolerances: the same per-output figures as tests/test_gpu_parity.py (TOL)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import make_inputs, relerr

FP = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return a.ctypes.data_as(FP) if a is not None else None


def _load(robot):
    import torch  # noqa: F401  (FIRST: torch carries its own HIP runtime; a process that loads /opt/rocm's copy before it -- as this
    #                harness would -- ends up with two runtimes and torch.cuda.is_available() turns False for the tests that follow)
    torch.cuda.is_available()
    from gridcodegenerator_amd import host
    path = host.build_api_harness(robot, host.DEFAULT_PRECISION)        # built by build(); rebuilt here only if stale
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    lib.as_run.restype = ctypes.c_int
    lib.as_run.argtypes = [ctypes.c_char_p, FP, FP, FP, ctypes.c_int, ctypes.c_float, FP]
    lib.as_last_error.restype = ctypes.c_char_p
    lib.as_num_joints.restype = ctypes.c_int
    return lib


def test_harness_library_exports():
    """CPU: the prebuilt harness of the small robots loads and exports its entry points (no compute without a GPU)."""
    from gridcodegenerator_amd import host
    for robot in ("iiwa7", "mixed5"):
        if not os.path.exists(host.api_harness_path(robot, host.DEFAULT_PRECISION)):
            pytest.skip("harness not built (run __graft_entry__.build())")
        lib = _load(robot)
        assert lib.as_num_joints() == host.get_robot(robot).get_num_joints()


HOST_VARIANTS = ["id_cmem", "id_qdd_cmem", "minv_cmem", "idgrad_cmem", "idgrad_qdd_cmem", "fdgrad_qddminv"]
CO_VARIANTS = ["id", "id_qdd", "id_cmem", "id_qdd_cmem", "minv", "minv_cmem", "fd", "idgrad", "idgrad_qdd", "idgrad_cmem",
               "idgrad_qdd_cmem", "fdgrad", "fdgrad_qddminv"]
LAUNCH_VARIANTS = ["id", "id_qdd_cmem", "minv", "minv_cmem", "fd", "idgrad", "idgrad_qdd_cmem", "fdgrad", "fdgrad_qddminv"]


@pytest.mark.gpu
def test_api_surface_on_gpu(robot_name, tables):
    from oracle import rbd_oracle as O
    from test_gpu_parity import TOL
    tol = TOL[robot_name]
    lib = _load(robot_name)
    n = lib.as_num_joints()
    K = 150                                     # two full tiles + a ragged one
    T = tables(robot_name)
    q, qd, u = make_inputs(n, K, 61)
    q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
    x = np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))
    df, parts = O.fd_grad(T, q64, qd64, u64, return_parts=True)
    gflat = lambda M: np.concatenate([O.flat_colmajor(M[:, :, :n]), O.flat_colmajor(M[:, :, n:])], axis=1)
    qdd32 = np.ascontiguousarray(parts["qdd"].astype(np.float32))
    Minv_ref = O.flat_colmajor(np.triu(parts["Minv"]))
    Minv32 = np.ascontiguousarray(Minv_ref.astype(np.float32))
    # ID with qdd: NOT the forward-dynamics result (ID(q, qd, FD(q, qd, u)) = u by cancellation, which measures nothing)
    qdd_alt32 = np.ascontiguousarray((0.7 * u).astype(np.float32))
    qdd_in = qdd_alt32.astype(np.float64)
    expect = {
        "id": (parts["c"], "c"), "id_qdd": (O.rnea(T, q64, qd64, qdd_in)[0], "c_qdd"), "minv": (Minv_ref, "Minv"), "fd": (parts["qdd"], "qdd"),
        "idgrad": (gflat(O.rnea_grad(T, q64, qd64, None)), "dc_du"), "idgrad_qdd": (gflat(O.rnea_grad(T, q64, qd64, qdd_in)), "dc_du"),
        "fdgrad": (gflat(df), "df_du"), "fdgrad_qddminv": (gflat(df), "df_du_qdd_minv"),
    }

    def run(name, count):
        out = np.full((K, count), np.nan, dtype=np.float32)
        qdd_arg = qdd32 if name.startswith("fdgrad") else qdd_alt32         # the qdd+Minv variant needs the matching qdd
        rc = lib.as_run(name.encode(), _p(x), _p(qdd_arg), _p(Minv32), K, ctypes.c_float(9.81), _p(out))
        assert rc == 0, (name, rc, lib.as_last_error().decode())
        return out

    worst = {}
    for (names, suffix) in ((HOST_VARIANTS, ""), (CO_VARIANTS, "_co"), (LAUNCH_VARIANTS, "_launch")):
        for v in names:
            ref, key = expect[v.replace("_cmem", "")]
            got = run(v + suffix, ref.shape[1])
            err = relerr(got, ref)[0]
            worst[v + suffix] = err
            assert err < tol[key], (v + suffix, err)
    # compressed-memory variants read the same numbers from a different buffer: bit-identical to the plain variants
    for a, b in (("id_co", "id_cmem_co"), ("id_qdd_co", "id_qdd_cmem_co"), ("minv_co", "minv_cmem_co"), ("idgrad_co", "idgrad_cmem_co"),
                 ("minv_launch", "minv_cmem_launch")):
        sz = expect[a.split("_co")[0].split("_launch")[0]][0].shape[1]
        assert np.array_equal(run(a, sz), run(b, sz)), (a, b)

    # ---- inverse_dynamics_vaf_device in a user kernel
    for name, qdd_arg in (("vaf_device", None), ("vaf_device_qdd", qdd_in)):
        c, v, a, f = O.rnea(T, q64, qd64, qdd_arg)
        ref = np.concatenate([v.reshape(K, 6 * n), a.reshape(K, 6 * n), f.reshape(K, 6 * n)], axis=1)
        assert relerr(run(name, 18 * n), ref)[0] < tol["c_qdd"], name

    # ---- the _device tier in a user kernel: [c | Minv | qdd | dc_du at the GPU's qdd]
    got = run("device_tier", 2 * n + 3 * n * n)
    assert relerr(got[:, :n], parts["c"])[0] < tol["c"]
    assert relerr(got[:, n:n + n * n], Minv_ref)[0] < tol["Minv"]
    qdd_gpu = got[:, n + n * n:2 * n + n * n]
    assert relerr(qdd_gpu, parts["qdd"])[0] < tol["qdd"]
    if n <= 12:
        assert relerr(got[:, 2 * n + n * n:], gflat(O.rnea_grad(T, q64, qd64, qdd_gpu.astype(np.float64))))[0] < tol["dc_du"]

    # ---- the _inner chain in a user kernel: [qdd | Minv | vaf at the GPU's qdd | dc_du at the GPU's qdd]
    got = run("inner_chain", n + n * n + 18 * n + 2 * n * n)
    qdd_gpu = got[:, :n]
    assert relerr(qdd_gpu, parts["qdd"])[0] < tol["qdd"]
    assert relerr(got[:, n:n + n * n], Minv_ref)[0] < tol["Minv"]
    c, v, a, f = O.rnea(T, q64, qd64, qdd_gpu.astype(np.float64))
    vaf = np.concatenate([v.reshape(K, 6 * n), a.reshape(K, 6 * n), f.reshape(K, 6 * n)], axis=1)
    assert relerr(got[:, n + n * n:n + n * n + 18 * n], vaf)[0] < tol["c_qdd"]
    assert relerr(got[:, n + n * n + 18 * n:], gflat(O.rnea_grad(T, q64, qd64, qdd_gpu.astype(np.float64))))[0] < tol["dc_du"]
    print("api surface %s: worst norm-wise errors %s" % (robot_name, {k: "%.1e" % v for k, v in sorted(worst.items())}))


@pytest.mark.gpu
def test_spatial_algebra_device_library_on_gpu():
    """The reference header's device library (helpers/_spatial_algebra_helpers.py:35-257) EXECUTED ON THE GPU from a user kernel
    (tests/api_surface_harness.hip: spatial_kernel, one lane per (x, y) pair): the runtime-selected mxX family (plain, _scaled, _peq,
    _peq_scaled), mx0..mx5 by name, fx, fx_zeroed, fx_times_v, fx_times_v_peq and dot_prod against the oracle's cross-product
    matrices -- mxK(x) = crm(x)[:, K], fx(x) = crf(x) = -crm(x)^T (column-major), fx_times_v(x, y) = crf(x) y."""
    from oracle import rbd_oracle as O
    lib = _load("iiwa7")
    lib.as_spatial.restype = ctypes.c_int
    lib.as_spatial.argtypes = [FP, ctypes.c_float, ctypes.c_int, FP]
    lib.as_spatial_row.restype = ctypes.c_int
    K, row = 130, lib.as_spatial_row()                       # (two full waves + two lanes: the kernel's tail guard is exercised)
    assert row == 6 * 4 * 6 + 36 + 36 + 6 + 6 + 2 + 36
    rng = np.random.default_rng(5)
    xy = rng.uniform(-2, 2, (K, 12)).astype(np.float32)
    alpha = np.float32(0.37)
    out = np.full((K, row), np.nan, dtype=np.float32)
    rc = lib.as_spatial(_p(xy), ctypes.c_float(alpha), K, _p(out))
    assert rc == 0, lib.as_last_error().decode()
    assert np.isfinite(out).all()
    for k in (0, 1, 63, 64, 129):
        x64, y64 = xy[k, :6].astype(np.float64), xy[k, 6:].astype(np.float64)
        crm = np.stack([O.mxS(c, x64) for c in range(6)], axis=1)
        crf = O.fx(x64)
        o = 0
        for c in range(6):
            col = crm[:, c]
            for expect in (col, col * alpha, y64 + col, y64 + col * alpha):
                np.testing.assert_allclose(out[k, o:o + 6], expect, rtol=1e-6, atol=1e-6)
                o += 6
        for _ in range(2):
            np.testing.assert_allclose(out[k, o:o + 36].reshape(6, 6).T, crf, rtol=1e-6, atol=1e-6)       # column-major
            o += 36
        np.testing.assert_allclose(out[k, o:o + 6], crf @ y64, rtol=1e-5, atol=1e-5); o += 6
        np.testing.assert_allclose(out[k, o:o + 6], y64 + crf @ y64, rtol=1e-5, atol=1e-5); o += 6
        np.testing.assert_allclose(out[k, o], x64 @ y64, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out[k, o + 1], x64[0] * y64[0] + x64[2] * y64[1] + x64[4] * y64[2], rtol=1e-5, atol=1e-6); o += 2
        np.testing.assert_allclose(out[k, o:o + 36].reshape(6, 6).T, crm, rtol=1e-6, atol=1e-6)           # mx0..mx5 by name = the columns of crm(x)
