"""Diagnostic: repeatability of the register-capped Atlas-30 kernels (tests/regression_variants.py: atlas30_capped) at K = 16384:
which output columns are non-finite between repeated launches of kernels with DIFFERENT scratch sizes.
usage: python tools/diag/capped_repeat.py <variant> <K> [warm]     (warm: first launch the kernel with the largest scratch once)"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from gridcodegenerator_amd import host
import regression_variants
regression_variants.register()
name = sys.argv[1] if len(sys.argv) > 1 else "atlas30_capped"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
warm = len(sys.argv) > 3 and sys.argv[3] == "warm"
h = host.GridHandle(name, precision="fp32"); n = h.n
x = np.random.default_rng(9).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
d_in = torch.from_numpy(x).cuda()
dID, dFD = host.ALG_ID_DU, host.ALG_FD_DU
h.set_coop(dFD, 1)
calls = {dID: h.inverse_dynamics_gradient_device, dFD: h.forward_dynamics_gradient_device}
def launch(alg, S, Kx=K):
    h.set_split(alg, S)
    out = torch.full((Kx, 2 * n * n), float("nan"), dtype=torch.float32, device="cuda")
    calls[alg](out.data_ptr(), d_in.data_ptr(), 3 * n, Kx)
    h.synchronize()
    o = out.cpu().numpy()
    bad = ~np.isfinite(o)
    cols = np.nonzero(bad.reshape(Kx, 2 * n, n).any(axis=(0, 2)))[0]
    scratch = h.L.kernel_attributes(alg, split=S if S > 1 else 0)["scratch_bytes_per_lane"]
    return int(bad.sum()), cols[:8].tolist(), scratch
if warm:
    print("warm-up: dFD unsplit (largest scratch) K=64:", launch(dFD, 1, 64), flush=True)
seq = [(dID, 1), (dID, 2), (dID, 4), (dID, 1), (dID, 1), (dFD, 4), (dID, 1), (dFD, 1), (dID, 4), (dID, 1), (dFD, 2), (dFD, 1), (dID, 1)]
total_bad = 0
for rep in range(3):
    row = []
    for alg, S in seq:
        nb, cols, scratch = launch(alg, S)
        total_bad += nb > 0
        row.append("%s/S%d[%dB]:%s" % ("dID" if alg == dID else "dFD", S, scratch, "ok" if nb == 0 else "BAD%s" % cols))
    print("rep %d: %s" % (rep, " ".join(row)), flush=True)
print("launches with non-finite output: %d of %d" % (total_bad, 3 * len(seq)))
h.close()
