"""Emitted header compiled for the HOST and run on the CPU (tests/host_harness.cpp): checks the generated
C++ text -- cores, _device wrappers and the pointer-style _inner tier -- against the oracle without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import make_inputs, relerr
from gridcodegenerator_amd import host
from oracle import rbd_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FP = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return a.ctypes.data_as(FP)


@pytest.fixture(scope="module", params=["iiwa7", "mixed5"])
def harness(request, tmp_path_factory, robots):
    name = request.param
    d = tmp_path_factory.mktemp("hh_" + name)
    header = str(d / ("grid_%s.hip.h" % name))
    host.generate_header(robots(name), header, "grid_" + name)
    so = str(d / "libhh.so")
    cmd = [host._hipcc(), "--offload-host-only", "-O2", "-march=native", "-fPIC", "-shared", "-std=c++17",
           "-DGRID_HEADER=\"%s\"" % header, "-DGRID_NS=grid_" + name, "-x", "hip", os.path.join(HERE, "host_harness.cpp"), "-o", so]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout[-3000:]
    return name, ctypes.CDLL(so)


def test_host_compiled_header(harness, tables):
    name, lib = harness
    T = tables(name)
    n = lib.hh_num_joints()
    K = 16
    q, qd, u = make_inputs(n, K, 21)
    x = np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))
    q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
    g = ctypes.c_float(9.81)

    ref_df, parts = O.fd_grad(T, q64, qd64, u64, return_parts=True)
    ref_df = np.concatenate([O.flat_colmajor(ref_df[:, :, :n]), O.flat_colmajor(ref_df[:, :, n:])], axis=1)
    out = np.zeros((K, 2 * n * n), dtype=np.float32)
    lib.hh_fd_grad_f64(_p(x), _p(out), K, g)
    assert relerr(out, ref_df)[0] < 2e-7          # fp64 arithmetic, fp32 rounding of the result only
    lib.hh_fd_grad_f32(_p(x), _p(out), K, g)
    assert relerr(out, ref_df)[0] < 1e-5          # fp32 arithmetic

    c = np.zeros((K, n), dtype=np.float32)
    lib.hh_id(_p(x), None, _p(c), K, g)
    assert relerr(c, O.rnea(T, q64, qd64)[0])[0] < 2e-6
    qdd = np.ascontiguousarray((0.7 * u).astype(np.float32))   # (qdd = FD(u) would make c = u by cancellation)
    lib.hh_id(_p(x), _p(qdd), _p(c), K, g)
    assert relerr(c, O.rnea(T, q64, qd64, qdd.astype(np.float64))[0])[0] < 2e-6

    Mi = np.zeros((K, n * n), dtype=np.float32)
    lib.hh_minv(_p(x), _p(Mi), K)
    assert relerr(Mi, O.flat_colmajor(O.minv(T, q64, False)))[0] < 5e-6

    acc = np.zeros((K, n), dtype=np.float32)
    lib.hh_fd(_p(x), _p(acc), K, g)
    assert relerr(acc, parts["qdd"])[0] < 2e-5

    # _inner tier, chained like the reference's fused kernel
    qdd_o = np.zeros((K, n), dtype=np.float32); dc = np.zeros((K, 2 * n * n), dtype=np.float32)
    lib.hh_inner_chain(_p(x), _p(qdd_o), _p(dc), K, g)
    assert relerr(qdd_o, parts["qdd"])[0] < 2e-5
    ref_dc = O.rnea_grad(T, q64, qd64, qdd_o.astype(np.float64))
    ref_dc = np.concatenate([O.flat_colmajor(ref_dc[:, :, :n]), O.flat_colmajor(ref_dc[:, :, n:])], axis=1)
    assert relerr(dc, ref_dc)[0] < 5e-6


def test_inline_sincos_accuracy(harness):
    """The emitted grid_sincos (branch-free Cody-Waite + minimax) against float64 sin/cos on its stated domain |x| <= 1e6, its
    graceful degradation just beyond (1e7), and NaN for non-finite input like the library."""
    name, lib = harness
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-1e5, 1e5, 200000), rng.uniform(-1e6, 1e6, 200000),
                        [0.0, -0.0, np.pi / 2, -np.pi / 2, np.pi, 1e-30, 1e6, -1e6]]).astype(np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    lib.hh_sincos(_p(x), _p(s), _p(c), len(x))
    x64 = x.astype(np.float64)
    assert np.abs(s - np.sin(x64)).max() < 1.5e-7 and np.abs(c - np.cos(x64)).max() < 1.5e-7
    x = rng.uniform(-1e7, 1e7, 20000).astype(np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    lib.hh_sincos(_p(x), _p(s), _p(c), len(x))
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 3e-4 and np.abs(c - np.cos(x.astype(np.float64))).max() < 3e-4
    x = np.array([np.inf, -np.inf, np.nan], dtype=np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    lib.hh_sincos(_p(x), _p(s), _p(c), len(x))
    assert np.isnan(s).all() and np.isnan(c).all()


def test_spatial_algebra_device_library(harness):
    """mx0..5 (+ _peq, _scaled, _peq_scaled, through the runtime-selected mxX family), fx, fx_zeroed, fx_times_v(_peq), dot_prod --
    the reference header's device library (helpers/_spatial_algebra_helpers.py:35-257) -- against the oracle's cross-product
    matrices: mxK(x) = crm(x)[:, K], fx(x) = crf(x) = -crm(x)^T (column-major), fx_times_v(x, y) = crf(x) y."""
    name, lib = harness
    rng = np.random.default_rng(5)
    x = rng.uniform(-2, 2, 6).astype(np.float32); y = rng.uniform(-2, 2, 6).astype(np.float32)
    alpha = np.float32(0.37)
    out = np.zeros(6 * 4 * 6 + 36 + 36 + 6 + 6 + 2, dtype=np.float32)
    lib.hh_spatial(_p(x), _p(y), ctypes.c_float(alpha), _p(out))
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    crm = np.stack([O.mxS(k, x64) for k in range(6)], axis=1)       # column K of crm(x) = x x e_K
    crf = O.fx(x64)
    np.testing.assert_allclose(crf, -crm.T, atol=0)                   # the two oracle functions agree with the identity crf = -crm^T
    o = 0
    for k in range(6):
        col = crm[:, k]
        for expect in (col, col * alpha, y64 + col, y64 + col * alpha):
            np.testing.assert_allclose(out[o:o + 6], expect, rtol=1e-6, atol=1e-6)
            o += 6
    for _ in range(2):
        np.testing.assert_allclose(out[o:o + 36].reshape(6, 6).T, crf, rtol=1e-6, atol=1e-6)       # column-major
        o += 36
    np.testing.assert_allclose(out[o:o + 6], crf @ y64, rtol=1e-5, atol=1e-5); o += 6
    np.testing.assert_allclose(out[o:o + 6], y64 + crf @ y64, rtol=1e-5, atol=1e-5); o += 6
    np.testing.assert_allclose(out[o], x64 @ y64, rtol=1e-5)
    np.testing.assert_allclose(out[o + 1], x64[0] * y64[0] + x64[2] * y64[1] + x64[4] * y64[2], rtol=1e-5)


def test_launch_shape_sanitiser_folds_z(harness):
    """The dim3 host wrappers number threads as x + y*size_x (the reference's kernels: helpers/_code_generation_helpers.py:41-55); a z
    extent would hand several waves the same thread id -- and the same LDS staging region.  grid_launch_dims + grid_fold_launch_z:
    illegal shapes become the suggested one, z extents are folded into y with the thread / block count unchanged."""
    name, lib = harness
    shape = lambda *v: (ctypes.c_int * 6)(*v)
    io = shape(4, 1, 1, 64, 2, 2)                      # 256 threads as 64 x 2 x 2
    lib.hh_launch_shape(io, 1000)
    assert list(io) == [4, 1, 1, 64, 4, 1]
    io = shape(2, 3, 2, 128, 1, 1)                     # 12 blocks as 2 x 3 x 2
    lib.hh_launch_shape(io, 1000)
    assert list(io) == [2, 6, 1, 128, 1, 1]
    io = shape(1, 1, 1, 64, 4, 2)                      # 512 threads: more than the kernels accept -> the suggested shape
    lib.hh_launch_shape(io, 1000)
    assert io[2] == 1 and io[5] == 1 and io[3] * io[4] <= 256 and io[0] * io[3] * io[4] >= 1000 // 1 or io[0] >= 1
    io = shape(0, 0, 0, 0, 0, 0)
    lib.hh_launch_shape(io, 100)
    assert io[0] >= 1 and io[3] >= 64 and io[2] == 1 and io[5] == 1
