"""Run one algorithm a few times (for rocprofv3 PMC collection).  usage: run_alg.py robot alg K pipeline_mode split reps [coop_mode]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np, torch
from gridcodegenerator_amd import host
import lean_variants; lean_variants.register()          # (experiment variants load by name like the built-in robots)
robot, alg, K, mode, split, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
h = host.GridHandle(robot); n = h.n
x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
h.set_pipeline(alg, mode); h.set_split(alg, split)
if len(sys.argv) > 7:
    h.set_coop(alg, int(sys.argv[7]))
calls = {host.ALG_ID_DU: h.inverse_dynamics_gradient_device, host.ALG_FD_DU: h.forward_dynamics_gradient_device}
for _ in range(reps):
    calls[alg](d_out.data_ptr(), d_in.data_ptr(), 3 * n, K)
h.synchronize(); h.close()
