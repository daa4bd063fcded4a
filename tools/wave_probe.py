"""GPU box: the wave-per-configuration kernels of a library (forced) over batch sizes, all five algorithms, next to another build of the
same robot.  usage: python tools/wave_probe.py K,K,... name [name ...]   (names: built-in robots or tools/lean_variants.py variants)"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np, torch
from gridcodegenerator_amd import host
import lean_variants
lean_variants.register()
Ks = [int(x) for x in sys.argv[1].split(',')]
for name in sys.argv[2:]:
    h = host.GridHandle(name, precision="fp32"); n = h.n
    for alg in range(5):
        h.set_wave(alg, 2)
        a = h.L.kernel_attributes(alg, wave=True)
        row = []
        for K in Ks:
            x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
            d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((K, host.output_size(alg, n)), dtype=torch.float32, device='cuda')
            h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=100)
            us = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=100) for _ in range(3)) * 1e3
            row.append("K=%d %6.2f" % (K, us))
        print("%-16s %-28s regs %3d scratch %3d B | %s" % (name, host.ALG_NAMES[alg], a["numRegs"], a["scratch_bytes_per_lane"], " | ".join(row)), flush=True)
    h.close()
