"""Atlas-30 in the MIXED arithmetic: the register-lean 8-wave forward-dynamics-gradient kernel (double Minv passes and qdd rows inside the
waves, float across LDS; on request: grid_set_coop mode 3) against the library's automatic choice (the 4-wave kernel: double
recursion kept in registers end to end) and against the fp32 library's lean kernel -- error against the oracle on the batches of the
precision report, and time per launch.  Run on the GPU box: `python tests/gpu_checks/mixed_lean_report.py > gpurun_out/...`."""
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
    robot, alg = "atlas30", host.ALG_FD_DU
    T = O.RobotTables(get_robot(robot))
    # the lean block in the mixed arithmetic is not in the shipped mixed library: build it as a variant first
    #   host.register_variant("atlas30_lm", "atlas30", share_objects=True, experimental={"lean_mixed": True}); host.build_library("atlas30_lm", "mixed")
    host.register_variant("atlas30_lm", "atlas30", share_objects=True, experimental={"lean_mixed": True})
    handles = {"fp32 lean (automatic)": (host.GridHandle(robot, precision="fp32"), 0),
               "mixed 4-wave (automatic)": (host.GridHandle(robot, precision="mixed"), 0),
               "mixed lean (mode 3)": (host.GridHandle("atlas30_lm", precision="mixed"), 3)}
    n = handles["fp32 lean (automatic)"][0].n
    for name, (h, mode) in handles.items():
        h.set_coop(alg, mode); h.set_wave(alg, 1)
        a = h.L.kernel_attributes(alg, coop=h.get_coop(alg, 16384))
        print("%-26s registers %d, scratch %d B per lane" % (name, a["numRegs"], a["scratch_bytes_per_lane"]))
    print("norm-wise error of df_du against the oracle (max|err| / max|ref|)")
    for (K, seed) in ((201, 31), (333, 47), (2048, 31), (2048, 5), (2048, 32)):
        q, qd, u = make_inputs(n, K, seed)
        ref = oracle_all(T, q, qd, u)["df_du"]
        d_in = torch.from_numpy(pack(q, qd, u)).cuda()
        row = []
        for name, (h, mode) in handles.items():
            out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
            h.forward_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
            row.append("%s %.2e" % (name, relerr(out.cpu().numpy(), ref)[0]))
        print("K=%-5d seed %-3d | %s" % (K, seed, " | ".join(row)), flush=True)
    print("time per launch [us]")
    for K in (64, 4096, 16384, 65536):
        rng = np.random.default_rng(0)
        x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
        d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
        reps = max(3, min(200, int(4e6 / K)))
        row = []
        for name, (h, mode) in handles.items():
            h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)
            us = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(3)) * 1e3
            row.append("%s %8.2f" % (name, us))
        print("K=%-6d | %s" % (K, " | ".join(row)), flush=True)
    # the inverse-dynamics gradient of the mixed library is the fp32 arithmetic (no double part): the lean kernel is automatic there too
    hm = handles["mixed 4-wave (automatic)"][0]
    print("mixed library, inverse-dynamics gradient: grid_get_coop(K = 16384) =", hm.get_coop(host.ALG_ID_DU, 16384))
    for name, (h, mode) in handles.items():
        h.close()


if __name__ == "__main__":
    main()
