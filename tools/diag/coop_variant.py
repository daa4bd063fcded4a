"""Diagnostic: the tile-cooperative kernel of a variant library against the shipped one -- same numbers? how fast?
usage: python3 tools/diag/coop_variant.py <variant.so>"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from gridcodegenerator_amd import host
path = sys.argv[1]
libs = {"shipped": host.GridLibrary("atlas30", "fp32"), "variant": host.GridLibrary("atlas30", "fp32", path=path)}
hs = {k: host.GridHandle("atlas30", library=L) for k, L in libs.items()}
n = 30; alg = host.ALG_FD_DU
for K in (64, 1000, 16384, 65536):
    x = np.random.default_rng(K).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    outs, row = {}, []
    for k, h in hs.items():
        h.set_coop(alg, 2)
        out = torch.zeros((K, 2 * n * n), dtype=torch.float32, device="cuda")
        h.forward_dynamics_gradient_device(out.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
        outs[k] = out.cpu().numpy()
        reps = max(3, min(200, int(4e6 / K)))
        h.time_device(alg, out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps)
        t = min(h.time_device(alg, out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=reps) for _ in range(3)) * 1e3
        row.append("%s %7.2f us (scratch %d B)" % (k, t, h.L.kernel_attributes(alg, coop=True)["scratch_bytes_per_lane"]))
    same = np.array_equal(outs["shipped"], outs["variant"])
    print("K=%-6d %s | bitwise equal: %s (max |diff| %.2e)" % (K, " | ".join(row), same, np.abs(outs["shipped"] - outs["variant"]).max()), flush=True)
