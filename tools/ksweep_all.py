"""GPU box: time per launch of every algorithm of a robot over the batch size (automatic launch policy), to separate what a lone
tile costs from what a full chip costs.  usage: python tools/ksweep_all.py <robot> <precision> [K,K,...]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot, precision = sys.argv[1], sys.argv[2]
Ks = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else [64, 1024, 4096, 16384, 65536]
import os
if not os.environ.get('GRID_USE_PREBUILT'): host.build_library(robot, precision)
h = host.GridHandle(robot, precision=precision); n = h.n
algs = [("RNEA", host.ALG_ID, n), ("Minv", host.ALG_MINV, n * n), ("FD", host.ALG_FD, n), ("dID", host.ALG_ID_DU, 2 * n * n), ("dFD", host.ALG_FD_DU, 2 * n * n)]
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    row = []
    for (name, alg, cnt) in algs:
        d_out = torch.empty((K, cnt), dtype=torch.float32, device='cuda')
        h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=50)
        t = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=100) for _ in range(3)) * 1e3
        row.append("%s %8.2f us" % (name, t))
    print("%s %s K=%-6d | %s" % (robot, precision, K, " | ".join(row)), flush=True)
h.close()
