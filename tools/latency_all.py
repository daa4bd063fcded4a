"""GPU box: small-batch latency of all five algorithms: the automatic lane-per-configuration dispatch (wave kernels off) beside the
wave-per-configuration kernel, back-to-back launches on one stream, time per launch -- the measurement behind the
<ALG>_WAVE_AUTO_MAX_K constants of the generated header.  usage: python tools/latency_all.py <robot> <precision>"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot, precision = sys.argv[1], sys.argv[2]
host.build_library(robot, precision)
h = host.GridHandle(robot, precision=precision); n = h.n
NAMES = {host.ALG_ID: "ID", host.ALG_MINV: "MINV", host.ALG_FD: "FD", host.ALG_ID_DU: "ID_DU", host.ALG_FD_DU: "FD_DU"}
for K in (1, 64, 256, 512, 1024, 2048, 4096):
    x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    row = []
    for alg, name in NAMES.items():
        d_out = torch.empty((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
        def t():
            h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300)
            return min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300) for _ in range(3)) * 1e3
        h.set_wave(alg, 1); a = t()
        h.set_wave(alg, 2); b = t()
        h.set_wave(alg, 0)
        row.append("%s %6.2f / %6.2f" % (name, a, b))
    print("%s %s K=%-5d lanes / wave [us] | %s" % (robot, precision, K, " | ".join(row)), flush=True)
h.close()
