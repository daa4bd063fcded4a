"""Experiment: the register-lean 8-wave kernels (made for large robots) generated for a SMALL robot -- iiwa-7, one chain of 7 joints,
14 gradient half-columns over 8 waves -- against its shipped kernels (4-way half-column split), back to back on one box.
usage: python tools/exp_small_lean.py build | run [K,K,...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gridcodegenerator_amd import host

VARIANTS = {"iiwa7_lean": ("iiwa7", dict(experimental={"lean_min_joints": 1})),
            "quad12_lean": ("quad12", dict(experimental={"lean_min_joints": 1}))}
for name, (base, kw) in VARIANTS.items():
    host.register_variant(name, base, share_objects=True, **kw)

if sys.argv[1] == "build":
    for name in VARIANTS:
        path = host.build_library(name, "fp32", verbose=True)
        print(name, [(k["name"], k.get("vgprs"), k.get("scratch")) for k in host.kernel_resources(name, "fp32") if k["name"].endswith("coop8")], flush=True)
else:
    import numpy as np, torch
    Ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,4096,16384,65536").split(",")]
    for name, (base, _) in VARIANTS.items():
        h = host.GridHandle(name, precision="fp32")
        n = h.n
        for alg, label in ((host.ALG_FD_DU, "dFD"), (host.ALG_ID_DU, "dID")):
            for K in Ks:
                rng = np.random.default_rng(0)
                x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
                d_in = torch.from_numpy(x).cuda()
                outs, row = {}, []
                for mode, what in ((1, "shipped"), (3, "lean")):
                    h.set_coop(alg, mode); h.set_wave(alg, 1)
                    d_out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
                    reps = max(5, min(300, int(4e6 / K)))
                    h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)
                    us = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(4)) * 1e3
                    outs[what] = d_out.cpu().numpy()
                    row.append("%s %7.2f us" % (what, us))
                err = np.abs(outs["lean"] - outs["shipped"]).max() / np.abs(outs["shipped"]).max()
                a = h.L.kernel_attributes(alg, coop=2)
                print("%-12s %s K=%6d | %s | max|diff|/max %.1e | lean kernel: %d registers, %d B scratch" % (name, label, K, " | ".join(row), err, a["numRegs"], a["scratch_bytes_per_lane"]), flush=True)
        h.close()
