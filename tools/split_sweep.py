"""Kernel time vs column split and batch for one gradient algorithm.  usage: split_sweep.py robot alg K1 K2 ..."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot = sys.argv[1]; alg = int(sys.argv[2]); Ks = [int(a) for a in sys.argv[3:]]
h = host.GridHandle(robot); n = h.n
splits = [1] + list(h.L.splits(alg))
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
    ref = None; row = []
    for S in splits:
        h.set_split(alg, S)
        ms = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=20) for _ in range(3))
        o = d_out.clone()
        if ref is None: ref = o
        row.append('S=%d %8.1f us (maxdiff %.1e)' % (S, ms * 1e3, float((o - ref).abs().max())))
    print('%s %s K=%d  %s' % (robot, host.ALG_NAMES[alg], K, ' | '.join(row)), flush=True)
h.close()
