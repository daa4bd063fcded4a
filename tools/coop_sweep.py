"""GPU box: forward-dynamics-gradient kernel time per variant (unsplit, automatic column split, tile-cooperative) over batch sizes.
usage: python tools/coop_sweep.py <robot> <precision> K,K,..."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot, precision = sys.argv[1], sys.argv[2]
Ks = [int(x) for x in sys.argv[3].split(',')]
import os
if not os.environ.get('GRID_USE_PREBUILT'): host.build_library(robot, precision)
h = host.GridHandle(robot, precision=precision); n = h.n
alg = host.ALG_FD_DU
for a in ("unsplit", "coop"):
    pass
for K in Ks:
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
    reps = max(3, min(200, int(4e6 / K)))
    row = []
    def t():
        h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)     # ramp
        return min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(3)) * 1e3
    h.set_wave(alg, 1)
    h.set_coop(alg, 1); h.set_split(alg, 1); row.append("unsplit %8.2f us" % t())
    h.set_split(alg, 0); row.append("auto-split(S=%d) %8.2f us" % (h.get_split(alg, K), t()))
    if h.coop_available(alg):
        h.set_coop(alg, 2); us = t(); row.append("coop %8.2f us (%.3g evals/s)" % (us, K / us * 1e6))
        h.set_coop(alg, 0)
    if h.wave_available(alg) and K <= 8192:
        h.set_wave(alg, 2); us = t(); row.append("wave %8.2f us" % us); h.set_wave(alg, 0)
    print("%s %s K=%-8d | %s" % (robot, precision, K, " | ".join(row)), flush=True)
for name, kw in (("unsplit", {}), ("coop", {"coop": True}), ("wave", {"wave": True})):
    try:
        print(name, h.L.kernel_attributes(alg, **kw))
    except Exception as e:
        print(name, e)
h.close()
