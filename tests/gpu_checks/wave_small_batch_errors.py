"""Measured norm-wise errors of the wave-per-configuration kernels at the batch sizes tests/test_wave.py uses (K = 1, 7, 64, 200):
the figures behind WAVE_TOL there (every tolerance <= 3x what this prints).  Same seeds as the tests, plus four more seeds per K so
that the bound is not tuned to one sample.  Run on the GPU box: `python tests/gpu_checks/wave_small_batch_errors.py > gpurun_out/...`.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch
    from conftest import make_inputs, relerr
    from gridcodegenerator_amd import host
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    from test_gpu_parity import oracle_all, pack
    Ks = (1, 7, 64, 200)
    print("norm-wise error max|err|/max|ref| of the wave-per-configuration kernels, %s arithmetic; worst over seeds (test seed first)" % host.DEFAULT_PRECISION)
    for robot in ("iiwa7", "mixed5", "quad12", "atlas30"):
        T = O.RobotTables(get_robot(robot))
        worst = {}
        with host.GridHandle(robot, device=0, precision=host.DEFAULT_PRECISION) as h:
            n = h.n
            for a in range(5):
                h.set_wave(a, 2)
            for K in Ks:
                for rep, seed in enumerate([60 + K, 40 + K, 1000 + K, 2000 + K, 3000 + K, 4000 + K]):
                    q, qd, u = make_inputs(n, K, seed)
                    ref = oracle_all(T, q, qd, u)
                    d_in = torch.from_numpy(pack(q, qd, u)).cuda()
                    qdd_alt = np.random.default_rng(70 + K + rep).uniform(-1.0, 1.0, (K, n)).astype(np.float32)
                    d_qdd = torch.from_numpy(qdd_alt).cuda()
                    q64, qd64, qdd64 = (x.astype(np.float64) for x in (q, qd, qdd_alt))
                    ref["c_qdd"] = O.rnea(T, q64, qd64, qdd64)[0]
                    dc = O.rnea_grad(T, q64, qd64, qdd64)
                    ref["dc_du_qdd"] = np.concatenate([O.flat_colmajor(dc[:, :, :n]), O.flat_colmajor(dc[:, :, n:])], axis=1)
                    mk = lambda cols: torch.zeros((K, cols), dtype=torch.float32, device="cuda")
                    res = {}
                    o = mk(n); h.inverse_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K); res["c"] = o
                    o = mk(n); h.inverse_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, d_qdd=d_qdd.data_ptr()); res["c_qdd"] = o
                    o = mk(n * n); h.direct_minv_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K); res["Minv"] = o
                    o = mk(n); h.forward_dynamics_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K); res["qdd"] = o
                    o = mk(2 * n * n); h.inverse_dynamics_gradient_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K); res["dc_du_noqdd"] = o
                    o = mk(2 * n * n); h.inverse_dynamics_gradient_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K, d_qdd=d_qdd.data_ptr()); res["dc_du_qdd"] = o
                    o = mk(2 * n * n); h.forward_dynamics_gradient_device(o.data_ptr(), d_in.data_ptr(), 3 * n, K); res["df_du"] = o
                    h.synchronize()
                    for key, v in res.items():
                        e = relerr(v.cpu().numpy(), ref[key])[0]
                        worst.setdefault((key, K), []).append(e)
        for key in ("c", "c_qdd", "Minv", "qdd", "dc_du_noqdd", "dc_du_qdd", "df_du"):
            print("%-8s %-12s " % (robot, key) + "  ".join("K=%-3d max %.2e (test seeds %.2e %.2e)" % (K, max(worst[(key, K)]), worst[(key, K)][0], worst[(key, K)][1]) for K in Ks))


if __name__ == "__main__":
    main()
