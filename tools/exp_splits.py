"""Experiment: kernel time of the iiwa-7 FD-gradient for each column-split factor / block shape (GPU box)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
splits = [2, 3, 4, 5, 7]
host.build_library('iiwa7', 'fp32', force=True, grad_splits=splits)
h = host.GridHandle('iiwa7'); n = h.n
for K in (16384, 65536):
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
    for S in [1] + splits:
        h.set_split(host.ALG_FD_DU, S)
        row = []
        for threads in (64, 128, 256):
            ms = min(h.time_device(host.ALG_FD_DU, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, threads=threads, reps=200) for _ in range(3))
            row.append('%d thr: %6.2f us' % (threads, ms * 1e3))
        print('K=%d S=%d' % (K, S), ' | '.join(row), flush=True)
