"""Experiment (GPU box): kernel times of one robot built with extra hipcc flags.  usage: exp_flags.py robot "<flags>" K,K,... [alg]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot, flags = sys.argv[1], tuple(f for f in sys.argv[2].split() if f)
Ks = [int(x) for x in sys.argv[3].split(',')]
alg = int(sys.argv[4]) if len(sys.argv) > 4 else host.ALG_FD_DU
host.build_library(robot, 'fp32', force=True, extra_flags=flags)
h = host.GridHandle(robot); n = h.n
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
    row = []
    for S in [1] + h.L.splits(alg):
        h.set_split(alg, S)
        ms = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=200) for _ in range(3))
        row.append('S=%d %7.2f us' % (S, ms * 1e3))
    print('%s flags=%r K=%d |' % (robot, ' '.join(flags), K), ' | '.join(row), flush=True)
h.close()
