"""Diagnostic: many alternating launches of the register-capped Atlas-30 gradient kernels into NaN-prefilled buffers; for every
launch with non-finite output: which columns / how many rows.  mode "nosync": prefill and launch back to back on the null stream;
mode "sync": torch.cuda.synchronize() between the prefill and the launch; mode "cur": launch on torch.cuda.current_stream()
(its handle is printed); mode "own": prefill and launch on one torch.cuda.Stream() created here.
usage: python tools/diag/capped_stress.py <variant> <K> nosync|sync <launches>"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from gridcodegenerator_amd import host
import regression_variants
regression_variants.register()
name, K, mode, N = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
precision = sys.argv[5] if len(sys.argv) > 5 else "fp32"
h = host.GridHandle(name, precision=precision); n = h.n
x = np.random.default_rng(9).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
d_in = torch.from_numpy(x).cuda()
dID, dFD = host.ALG_ID_DU, host.ALG_FD_DU
calls = {dID: h.inverse_dynamics_gradient_device, dFD: h.forward_dynamics_gradient_device}
seq = [(dID, 1), (dID, 4), (dFD, 1), (dID, 2), (dFD, 4), (dID, 1), (dFD, 2)] + ([(dFD, -1)] if h.coop_available(dFD) else [])     # -1: tile-cooperative
bad_launches = 0
out = torch.empty((K, 2 * n * n), dtype=torch.float32, device="cuda")
print("torch.cuda.current_stream().cuda_stream = %#x, default_stream = %#x" % (torch.cuda.current_stream().cuda_stream, torch.cuda.default_stream().cuda_stream), flush=True)
own = torch.cuda.Stream() if mode == "own" else None
torch.cuda.synchronize()
for i in range(N):
    alg, S = seq[i % len(seq)]
    if alg == dFD and h.coop_available(dFD):
        h.set_coop(dFD, 2 if S < 0 else 1)
    h.set_split(alg, max(S, 0))
    if own is not None:
        with torch.cuda.stream(own):
            out.fill_(float("nan"))
        calls[alg](out.data_ptr(), d_in.data_ptr(), 3 * n, K, stream=own.cuda_stream)
        own.synchronize()
    else:
        out.fill_(float("nan"))
        if mode == "sync":
            torch.cuda.synchronize()
        if mode == "cur":
            calls[alg](out.data_ptr(), d_in.data_ptr(), 3 * n, K, stream=torch.cuda.current_stream().cuda_stream)
        else:
            calls[alg](out.data_ptr(), d_in.data_ptr(), 3 * n, K)
    h.synchronize(); torch.cuda.synchronize()
    nbad = int((~torch.isfinite(out)).sum().item())
    if nbad:
        bad_launches += 1
        o = out.cpu().numpy(); bad = ~np.isfinite(o)
        cols = np.nonzero(bad.reshape(K, 2 * n, n).any(axis=(0, 2)))[0]
        rows = np.nonzero(bad.any(axis=1))[0]
        print("launch %d %s S=%d: %d non-finite, columns %s, %d rows (first %s, last %s)" % (
            i, "dID" if alg == dID else "dFD", S, nbad, cols.tolist()[:10], len(rows), rows[:5].tolist(), rows[-3:].tolist()), flush=True)
print("%s %s: %d of %d launches had non-finite output" % (name, mode, bad_launches, N), flush=True)
h.close()
