"""Experiment: explicit 4-way column partitions of the iiwa-7 gradient kernels (non-contiguous column sets balance the arithmetic
better; they flush one column at a time).  `build` (here, no GPU) compiles the variants as robots of their own; `run` (GPU box)
times the 4-way split at K = 16384.  usage: python tools/exp_iiwa_sets.py build|run"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gridcodegenerator_amd import host, robots  # noqa: E402

VARIANTS = {
    "iiwa7_setsA": [[0, 4], [1], [2, 6], [3, 5]],       # arithmetic 2854-2937 per group (contiguous default: 2504-3580)
    "iiwa7_setsB": [[0, 3], [1], [2], [4, 5, 6]],       # one non-contiguous group only
    "iiwa7_setsC": [[0, 4], [1], [2], [3, 5, 6]],
}
for name, parts in VARIANTS.items():
    if name not in robots.REGISTERED_ROBOTS:
        robots.register_robot(name, lambda: robots.get_robot("iiwa7"))
    host.DEFAULT_GEN_KWARGS[name] = dict(grad_splits=[parts])
# the generator's own exhaustive search over column sets (experimental split_sets), per algorithm
VARIANTS["iiwa7_setsAuto"] = "optimal_column_sets per algorithm"
if "iiwa7_setsAuto" not in robots.REGISTERED_ROBOTS:
    robots.register_robot("iiwa7_setsAuto", lambda: robots.get_robot("iiwa7"))
host.DEFAULT_GEN_KWARGS["iiwa7_setsAuto"] = dict(grad_splits=[4], experimental={"split_sets": True})

if __name__ == "__main__":
    if sys.argv[1] == "build":
        for name in VARIANTS:
            print(host.build_library(name, "fp32"))
    else:
        import numpy as np, torch
        K = 16384
        for name in ["iiwa7"] + list(VARIANTS):
            h = host.GridHandle(name, precision="fp32"); n = h.n
            x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
            d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
            row = []
            for alg, nm in ((host.ALG_FD_DU, "dFD"), (host.ALG_ID_DU, "dID")):
                h.set_split(alg, 4)
                h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300)
                t = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=300) for _ in range(5)) * 1e3
                row.append("%s split4 %6.2f us" % (nm, t))
            print("%-12s %s | %s" % (name, VARIANTS.get(name, "default (contiguous)"), " | ".join(row)), flush=True)
            h.close()
