"""Kernel time vs batch for the gradient kernels of one robot (single-kernel variants).  usage: ksweep.py robot K1 K2 ..."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot = sys.argv[1]; Ks = [int(a) for a in sys.argv[2:]]
h = host.GridHandle(robot); n = h.n
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    for alg in range(5):
        d_out = torch.empty((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
        ms = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=20) for _ in range(3))
        print('%s K=%d %-28s %9.1f us  %8.1f M evals/s  alg %.0f GB/s' % (robot, K, host.ALG_NAMES[alg], ms * 1e3, K / ms / 1e3, host.algorithmic_bytes(alg, n) * K / ms / 1e6), flush=True)
h.close()
