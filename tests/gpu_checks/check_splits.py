"""Column-split kernels of one robot against the unsplit kernel and the oracle (small batch).  usage: python tests/gpu_checks/check_splits.py robot"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
from gridcodegenerator_amd.robots import get_robot
from oracle import rbd_oracle as O
robot = sys.argv[1]
h = host.GridHandle(robot); n = h.n
T = O.RobotTables(get_robot(robot))
K = 200
rng = np.random.default_rng(5)
q = rng.uniform(-3, 3, (K, n)).astype(np.float32); qd = rng.uniform(-1, 1, (K, n)).astype(np.float32); u = rng.uniform(-1, 1, (K, n)).astype(np.float32)
x = np.concatenate([q, qd, u], axis=1); d_in = torch.from_numpy(x).cuda()
q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
gflat = lambda M: np.concatenate([O.flat_colmajor(M[:, :, :n]), O.flat_colmajor(M[:, :, n:])], axis=1)
refs = {host.ALG_ID_DU: gflat(O.rnea_grad(T, q64, qd64, None)), host.ALG_FD_DU: gflat(O.fd_grad(T, q64, qd64, u64))}
calls = {host.ALG_ID_DU: h.inverse_dynamics_gradient_device, host.ALG_FD_DU: h.forward_dynamics_gradient_device}
ok = True
for alg in (host.ALG_ID_DU, host.ALG_FD_DU):
    base = None
    for S in [1] + list(h.L.splits(alg)):
        h.set_split(alg, S)
        out = torch.full((K + 1, 2 * n * n), 7.5, dtype=torch.float32, device='cuda')
        calls[alg](out.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
        o = out.cpu().numpy()
        err = np.abs(o[:K] - refs[alg]).max() / np.abs(refs[alg]).max()
        if base is None: base = o
        same = bool(np.array_equal(o, base))
        print(robot, host.ALG_NAMES[alg], 'S=%d' % S, 'err vs oracle %.2e' % err, 'bitwise == unsplit', same, 'guard row ok', bool(np.all(o[K] == 7.5)), flush=True)
        ok = ok and err < 3e-5 and bool(np.all(o[K] == 7.5))
h.close()
sys.exit(0 if ok else 1)
