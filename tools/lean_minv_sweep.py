"""GPU box: Atlas-30 direct Minv -- lane-per-configuration kernel against the register-lean 8-wave kernel (`direct_minv_kernel_coop8`)
and the wave-per-configuration kernel, back to back.  usage: python tools/lean_minv_sweep.py [robot] K,K,..."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
robot = sys.argv[1] if len(sys.argv) > 1 else "atlas30"
Ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,256,512,768,1024,2048,4096,16384,32768,65536,262144").split(',')]
alg = host.ALG_MINV
h = host.GridHandle(robot, precision="fp32")
n = h.n
a = h.L.kernel_attributes(alg, coop=2)
print("%s direct_minv_kernel_coop8: %d registers, %d B scratch per lane" % (robot, a["numRegs"], a["scratch_bytes_per_lane"]))
print("%8s %14s %14s %14s   %s" % ("K", "lanes", "lean 8 waves", "wave/cfg", "M evals/s (lean)"))
for K in Ks:
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((K, n * n), dtype=torch.float32, device='cuda')
    reps = max(5, min(300, int(4e6 / K)))
    res = []
    for (coop, wave) in ((1, 1), (3, 1), (1, 2)):
        if wave == 2 and K > 4096:
            res.append(float("nan")); continue
        h.set_coop(alg, coop); h.set_wave(alg, wave)
        h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)
        res.append(min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(3)) * 1e3)
    print("%8d %11.2f us %11.2f us %11.2f us   %.1f" % (K, res[0], res[1], res[2], K / res[1]), flush=True)
h.close()
