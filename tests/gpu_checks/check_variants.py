"""Correctness + timing of every gradient kernel variant of one robot (single-kernel and two-pass), small and full batch.
usage: python tests/gpu_checks/check_variants.py [robot] [Kbig]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
t0 = time.time()
def log(*a): print('[%.1fs]' % (time.time() - t0), *a, flush=True)
from gridcodegenerator_amd import host
import torch
from oracle import rbd_oracle as O
from gridcodegenerator_amd.robots import get_robot
robot = sys.argv[1] if len(sys.argv) > 1 else 'atlas30'
Kbig = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
h = host.GridHandle(robot); n = h.n; log('handle', h.L.kernel_attributes(host.ALG_FD_DU))
T = O.RobotTables(get_robot(robot))
K = 96
rng = np.random.default_rng(0)
q = rng.uniform(-3, 3, (K, n)).astype(np.float32); qd = rng.uniform(-1, 1, (K, n)).astype(np.float32); u = rng.uniform(-1, 1, (K, n)).astype(np.float32)
x = np.concatenate([q, qd, u], axis=1)
q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
df, parts = O.fd_grad(T, q64, qd64, u64, return_parts=True)
gflat = lambda M: np.concatenate([O.flat_colmajor(M[:, :, :n]), O.flat_colmajor(M[:, :, n:])], axis=1)
ref_fd = gflat(df)
qdd = parts['qdd'].astype(np.float32); Minv = O.flat_colmajor(np.triu(parts['Minv'])).astype(np.float32)
ref_id0 = gflat(O.rnea_grad(T, q64, qd64, None)); ref_idq = gflat(O.rnea_grad(T, q64, qd64, qdd.astype(np.float64)))
def err(d, ref): return float(np.abs(np.nan_to_num(d, nan=1e9) - ref).max() / np.abs(ref).max())
for mode in (1, 2):   # 1: single kernel, 2: two-pass
    h.set_pipeline(host.ALG_ID_DU, mode); h.set_pipeline(host.ALG_FD_DU, mode)
    log('mode', mode, 'id_du      err', err(h.inverse_dynamics_gradient(x), ref_id0))
    log('mode', mode, 'id_du(qdd) err', err(h.inverse_dynamics_gradient(x, qdd=qdd), ref_idq))
    log('mode', mode, 'fd_du      err', err(h.forward_dynamics_gradient(x), ref_fd))
log('fd_du(qdd,Minv) err', err(h.forward_dynamics_gradient(x, qdd=qdd, Minv=Minv), ref_fd))
K = Kbig
xb = np.random.default_rng(1).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
d_in = torch.from_numpy(xb).cuda(); d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
for alg, name in ((host.ALG_ID_DU, 'id_du'), (host.ALG_FD_DU, 'fd_du')):
    for mode in (1, 2):
        h.set_pipeline(alg, mode)
        ms = h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=10)
        log('K', K, name, 'mode', mode, '%.1f us' % (1e3 * ms), '%.1f M evals/s' % (K / ms / 1e3))
h.close(); log('done')
