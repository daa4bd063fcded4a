"""Experiment (GPU box): fused vs two-pass gradient kernels."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot = sys.argv[1]; Ks = [int(x) for x in sys.argv[2].split(',')]
host.build_library(robot)
h = host.GridHandle(robot); n = h.n
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    for alg in (host.ALG_ID_DU, host.ALG_FD_DU):
        d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
        row = []
        for mode in (1, 2):
            h.set_pipeline(alg, mode); h.set_split(alg, 1)
            ms = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=max(3, min(100, int(1e6 / K)))) for _ in range(3))
            row.append('%s %8.2f us (%.3g ev/s)' % ('fused' if mode == 1 else 'two-pass', ms * 1e3, K / ms * 1e3))
        print('%s K=%d %s |' % (robot, K, host.ALG_NAMES[alg]), ' | '.join(row), flush=True)
h.close()
