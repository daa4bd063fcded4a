"""GPU box: the headline kernel (iiwa-7 forward-dynamics gradient, K = 16384) of the experiment variants (tools/iiwa_variants.py) next to
the shipped library, back to back on one box: the shipped 4-way split, the variant's 4-way split, and its asymmetric 8-way split where
it has one.  usage: python tools/exp_iiwa_r04.py [K]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np, torch
from gridcodegenerator_amd import host
import iiwa_variants
names = iiwa_variants.register()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
alg = host.ALG_FD_DU
ref = None
for rnd in range(2):
    for name in ["iiwa7"] + names:
        try:
            h = host.GridHandle(name, precision="fp32")
        except Exception as e:
            print("%-16s not built" % name); continue
        n = h.n
        rng = np.random.default_rng(0)
        x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
        d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device='cuda')
        row = []
        for S in h.L.splits(alg):
            if S not in (4, 8):
                continue
            h.set_split(alg, S)
            h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300)
            us = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300) for _ in range(4)) * 1e3
            out = d_out.cpu().numpy()
            if ref is None:
                ref = out
            a = h.L.kernel_attributes(alg, split=S)
            row.append("split%d %6.2f us (regs %d scratch %d B, bitwise equal to shipped: %s)" % (S, us, a["numRegs"], a["scratch_bytes_per_lane"], np.array_equal(out, ref)))
        print("round %d %-16s K=%d | %s" % (rnd, name, K, " | ".join(row)), flush=True)
        h.close()
