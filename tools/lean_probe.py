"""GPU box: time the register-lean 8-wave kernel of the experiment variants (tools/lean_variants.py) next to the shipped library,
back to back on one box.  usage: python tools/lean_probe.py K,K,... [variant ...]"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np, torch
from gridcodegenerator_amd import host
import lean_variants
names = lean_variants.register()
Ks = [int(x) for x in sys.argv[1].split(',')]
todo = ["atlas30"] + [v for v in (sys.argv[2:] or names)]
alg = host.ALG_FD_DU
for name in todo:
    try:
        h = host.GridHandle(name, precision="fp32")
    except Exception as e:
        print("%-24s not built (%s)" % (name, str(e)[:60])); continue
    n = h.n
    h.set_wave(alg, 1); h.set_coop(alg, 3)
    a = h.L.kernel_attributes(alg, coop=2)
    row = []
    for K in Ks:
        rng = np.random.default_rng(0)
        x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
        d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device='cuda')
        reps = max(3, min(200, int(4e6 / K)))
        h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)
        us = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(3)) * 1e3
        row.append("K=%d %7.2f us" % (K, us))
    # outputs against the shipped library's (ragged batch, rows past the batch untouched)
    Kp = 1500 + 37
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, (Kp, n)), rng.uniform(-1, 1, (Kp, 2 * n))], axis=1).astype(np.float32)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.full((Kp + 3, 2 * n * n), -7.0, dtype=torch.float32, device='cuda')
    h.forward_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3 * n, Kp); h.synchronize()
    res = d_out.cpu().numpy()
    if name == "atlas30":
        ref_out = res
        check = "reference"
    else:
        err = np.abs(res[:Kp] - ref_out[:Kp]).max() / np.abs(ref_out[:Kp]).max()
        check = "max|diff|/max|ref| %.1e, untouched rows %s" % (err, bool((res[Kp:] == -7.0).all()))
    print("%-24s regs %3d scratch %4d B | %s | %s" % (name, a["numRegs"], a["scratch_bytes_per_lane"], " | ".join(row), check), flush=True)
    h.close()
