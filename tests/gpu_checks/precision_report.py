"""Measured error of every kernel output against the float64 oracle, per robot: norm-wise (max|err| / max|ref| over the
batch) and worst element-wise.  usage: python tests/gpu_checks/precision_report.py [precision] [robots...]"""
import json
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from conftest import elementwise_err, make_inputs, relerr
from gridcodegenerator_amd import host
from gridcodegenerator_amd.robots import get_robot
from oracle import rbd_oracle as O

precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
robots = sys.argv[2:] or ["iiwa7", "atlas30", "mixed5", "quad12"]
report = {}
for robot in robots:
    import os
    if not os.environ.get("GRID_USE_PREBUILT"):          # (experiments ship a prebuilt library that the build guard would refuse)
        host.build_library(robot, precision)
    h = host.GridHandle(robot, precision=precision)
    n = h.n
    T = O.RobotTables(get_robot(robot))
    for (K, seed) in ((201, 31), (2048, 77)):
        q, qd, u = make_inputs(n, K, seed)
        q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
        x = np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))
        df, parts = O.fd_grad(T, q64, qd64, u64, return_parts=True)
        gflat = lambda M: np.concatenate([O.flat_colmajor(M[:, :, :n]), O.flat_colmajor(M[:, :, n:])], axis=1)
        qdd32 = parts["qdd"].astype(np.float32)
        Minv32 = O.flat_colmajor(np.triu(parts["Minv"])).astype(np.float32)
        pairs = {
            "c": (h.inverse_dynamics(x), parts["c"]),
            "Minv": (h.direct_minv(x), O.flat_colmajor(np.triu(parts["Minv"]))),
            "qdd": (h.forward_dynamics(x), parts["qdd"]),
            "dc_du": (h.inverse_dynamics_gradient(x), gflat(O.rnea_grad(T, q64, qd64, None))),
            "dc_du_qdd": (h.inverse_dynamics_gradient(x, qdd=qdd32), gflat(O.rnea_grad(T, q64, qd64, qdd32.astype(np.float64)))),
            "df_du": (h.forward_dynamics_gradient(x), gflat(df)),
            "df_du_qdd_minv": (h.forward_dynamics_gradient(x, qdd=qdd32, Minv=Minv32), gflat(df)),
        }
        # per output: norm-wise / worst element-wise over entries above 1e-3 of the scale / worst element-wise over entries above 1e-6
        res = {k: relerr(g, r) + (elementwise_err(g, r, 1e-3),) for k, (g, r) in pairs.items()}
        report["%s:%d" % (robot, K)] = {k: [float(v[0]), float(v[2]), float(v[1])] for k, v in res.items()}
        print("%-8s K=%-5d %s" % (robot, K, "  ".join("%s %.2e/%.1e/%.1e" % (k, v[0], v[2], v[1]) for k, v in res.items())), flush=True)
    h.close()
print(json.dumps({"precision": precision, "errors_normwise_elementwise1e-3_elementwise1e-6": report}))
