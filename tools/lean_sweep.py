"""GPU box: Atlas-30 (or any large robot) forward-dynamics gradient, 4-wave tile-cooperative kernel against its register-lean 8-wave
variant over batch sizes, back to back on one box (HIP events, best of 3 x reps launches), with both kernels' resources and the
maximum deviation between their outputs.   usage: python tools/lean_sweep.py <robot> K,K,..."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot = sys.argv[1]
Ks = [int(x) for x in sys.argv[2].split(',')]
precision = "fp32"
h = host.GridHandle(robot, precision=precision); n = h.n
alg = host.ALG_FD_DU
h.set_wave(alg, 1)
print("4 waves per tile:", h.L.kernel_attributes(alg, coop=1))
print("8 waves per tile:", h.L.kernel_attributes(alg, coop=2))
for K in Ks:
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    outs = {}
    reps = max(3, min(200, int(4e6 / K)))
    row = []
    for name, mode in (("coop (4 waves)", 2), ("coop8 (8 waves, lean)", 3), ("coop (4 waves) again", 2), ("coop8 again", 3)):
        h.set_coop(alg, mode)
        d_out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device='cuda')
        h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)     # ramp
        us = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(3)) * 1e3
        outs[mode] = d_out[: min(K, 4096)].cpu().numpy()
        row.append("%s %8.2f us (%.3g evals/s, %.1f %% of HBM on 7560 B/eval)" % (name, us, K / us * 1e6, 100 * K * 7560 / (us * 1e-6) / 8e12))
    dev = np.abs(outs[2].astype(np.float64) - outs[3]).max() / np.abs(outs[2]).max()
    print("%s K=%-8d | %s | max deviation between the two %.2e" % (robot, K, " | ".join(row), dev), flush=True)
h.close()
