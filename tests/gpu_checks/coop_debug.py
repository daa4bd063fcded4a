"""Where do tile-cooperative results depend on the launch shape?  usage: python tests/gpu_checks/coop_debug.py [robot]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import make_inputs
from gridcodegenerator_amd import host
robot = sys.argv[1] if len(sys.argv) > 1 else "iiwa7"
h = host.GridHandle(robot); n = h.n
for K in (70, 200):
    q, qd, u = make_inputs(n, K, 90 + K)
    x = np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))
    d_in = torch.from_numpy(x).cuda()
    h.set_coop(host.ALG_FD_DU, 2)
    outs = {}
    for tag, blocks in (("b0", 0), ("b1", 1), ("b1again", 1), ("b2", 2), ("b0again", 0)):
        out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K, blocks=blocks)
        h.synchronize()
        outs[tag] = out.cpu().numpy()
    ref = outs["b0"]
    for tag, o in outs.items():
        d = np.abs(o - ref)
        rows = np.unique(np.where(d > 0)[0]); cols = np.unique(np.where(d > 0)[1])
        print(robot, "K", K, tag, "max abs diff vs b0 %.3e" % d.max(), "rel %.3e" % (d.max() / np.abs(ref).max()), "rows", rows[:12], "n_rows", len(rows), "cols", cols[:16], "n_cols", len(cols))
h.close()
