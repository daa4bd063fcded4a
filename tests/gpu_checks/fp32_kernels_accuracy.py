"""Atlas-30 forward-dynamics gradient in the fp32 arithmetic: norm-wise error against the oracle of the register-lean, 4-wave, 4-way split
and unsplit kernels on the same batches (profiles/r04/fp32_kernels_accuracy.txt).  Run on the GPU box."""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from conftest import make_inputs, relerr
from gridcodegenerator_amd import host
from gridcodegenerator_amd.robots import get_robot
from oracle import rbd_oracle as O
from test_gpu_parity import oracle_all, pack
T = O.RobotTables(get_robot("atlas30"))
h = host.GridHandle("atlas30", precision="fp32"); n = h.n; alg = host.ALG_FD_DU
h.set_wave(alg, 1)
for (K, seed) in ((201, 31), (333, 47), (2048, 31), (2048, 5), (2048, 32), (1500, 90)):
    q, qd, u = make_inputs(n, K, seed)
    ref = oracle_all(T, q, qd, u)["df_du"]
    d_in = torch.from_numpy(pack(q, qd, u)).cuda()
    row = []
    for name, mode, split in (("lean", 3, 0), ("4-wave", 2, 0), ("split4", 1, 4), ("unsplit", 1, 1)):
        h.set_coop(alg, mode); h.set_split(alg, split)
        out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
        row.append("%s %.2e" % (name, relerr(out.cpu().numpy(), ref)[0]))
    print("K=%-5d seed %-3d | %s" % (K, seed, " | ".join(row)), flush=True)
