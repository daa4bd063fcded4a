import sys, time; sys.path.insert(0, '.')
import numpy as np
t0 = time.time()
def log(*a): print('[%.1fs]' % (time.time() - t0), *a, flush=True)
from gridcodegenerator_amd import host
import torch
from oracle import rbd_oracle as O
from gridcodegenerator_amd.robots import get_robot
robot = sys.argv[1] if len(sys.argv) > 1 else 'atlas30'
h = host.GridHandle(robot); n = h.n; log('handle')
T = O.RobotTables(get_robot(robot))
for K in (64,):
    rng = np.random.default_rng(0)
    q = rng.uniform(-3, 3, (K, n)).astype(np.float32); qd = rng.uniform(-1, 1, (K, n)).astype(np.float32); u = rng.uniform(-1, 1, (K, n)).astype(np.float32)
    x = np.concatenate([q, qd, u], axis=1)
    df, parts = O.fd_grad(T, q.astype(np.float64), qd.astype(np.float64), u.astype(np.float64), return_parts=True)
    ref = np.concatenate([O.flat_colmajor(df[:, :, :n]), O.flat_colmajor(df[:, :, n:])], axis=1)
    qdd = parts['qdd'].astype(np.float32); Minv = O.flat_colmajor(np.triu(parts['Minv'])).astype(np.float32)
    log('K', K, 'calling qdd_minv variant')
    d = h.forward_dynamics_gradient(x, qdd=qdd, Minv=Minv)
    bad = ~np.isfinite(d)
    log('nan frac', bad.mean(), 'rows with nan', np.unique(np.where(bad)[0])[:20], 'cols with nan', len(np.unique(np.where(bad)[1])), np.unique(np.where(bad)[1])[:40])
    good = ~bad
    log('err on finite', (np.abs(d - ref)[good]).max() / np.abs(ref).max())
    errcol = np.abs(np.where(good, d - ref, 0)).max(axis=0) / np.abs(ref).max()
    log('cols with err>1e-3', np.where(errcol > 1e-3)[0][:60])
h.close(); log('done')
