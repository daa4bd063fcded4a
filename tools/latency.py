"""GPU box: small-batch latency of the forward-dynamics gradient: back-to-back launches on one
stream (each waits for the previous one), time per launch, for the single-wave kernel, the automatic column split and the
tile-cooperative kernel; beside them the reference-style `_single_timing` twin (one configuration evaluated repeatedly INSIDE
one kernel: tests/single_timing_harness.hip).  usage: python tools/latency.py <robot> <precision>"""
import subprocess, sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot, precision = sys.argv[1], sys.argv[2]
host.build_library(robot, precision)
h = host.GridHandle(robot, precision=precision); n = h.n
alg = host.ALG_FD_DU
for K in (1, 16, 64, 128, 256, 512, 1024, 2048, 4096):
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
    def t():
        h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300)
        return min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300) for _ in range(3)) * 1e3
    row = []
    h.set_wave(alg, 1)
    h.set_coop(alg, 1); h.set_split(alg, 1); row.append("single-wave %7.2f us" % t())
    h.set_split(alg, 0); row.append("column split x%d %7.2f us" % (h.get_split(alg, K), t()))
    if h.coop_available(alg):
        h.set_coop(alg, 2); row.append("tile-cooperative %7.2f us" % t()); h.set_coop(alg, 0)
    if h.wave_available(alg):
        h.set_wave(alg, 2); row.append("wave-per-configuration %7.2f us" % t()); h.set_wave(alg, 0)
    print("%s %s K=%-5d | %s" % (robot, precision, K, " | ".join(row)), flush=True)
h.close()
if robot == "iiwa7":
    exe = host.build_single_timing_harness(robot, precision)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
    print("\n".join(l for l in out.splitlines() if l.startswith("Single Call")))
