"""Experiment (GPU box): kernel times for ONE generator variant (one process per variant: a rebuilt .so at the same
path is not reloaded by dlopen).  usage: exp_variants.py <robot> '<json kwargs>' <K,K,...> [alg]"""
import sys, json; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot, kwargs = sys.argv[1], json.loads(sys.argv[2])
Ks = [int(x) for x in sys.argv[3].split(',')]
alg = int(sys.argv[4]) if len(sys.argv) > 4 else host.ALG_FD_DU
host.build_library(robot, 'fp32', force=True, **kwargs)
h = host.GridHandle(robot); n = h.n
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
    row = []
    for S in [1] + h.L.splits(alg):
        h.set_split(alg, S)
        ms = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=max(3, min(200, int(2e6 / K)))) for _ in range(3))
        row.append('S=%d %8.2f us (%.3g ev/s)' % (S, ms * 1e3, K / ms * 1e3))
    a = h.L.kernel_attributes(alg)
    print('%s %s alg=%d K=%d regs=%d scratch=%d |' % (robot, json.dumps(kwargs), alg, K, a['numRegs'], a['scratch_bytes_per_lane']), ' | '.join(row), flush=True)
h.close()
