"""Round-3 experiment (iiwa-7, K = 16384 and neighbours): (1) the column groups of a tile as the waves of ONE block (the default
launch shape of the split kernels: 64*S threads) against single-wave blocks spread over the chip (threads = 64: the round-2
placement); (2) arbitrary column SETS flushed once per half (grid_out_colset) against the contiguous groups.
`build` (here, no GPU) compiles the variant library; `run` (GPU box) times them.  usage: python tools/exp_iiwa_r03.py build|run"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gridcodegenerator_amd import host, robots  # noqa: E402

VARIANTS = {"iiwa7_cols": dict(experimental={"split_half_columns": False})}
for name, kw in VARIANTS.items():
    if name not in robots.REGISTERED_ROBOTS:
        robots.register_robot(name, lambda: robots.get_robot("iiwa7"))
    host.DEFAULT_GEN_KWARGS[name] = dict(kw)

if __name__ == "__main__":
    if sys.argv[1] == "build":
        for name in ["iiwa7"] + list(VARIANTS):
            print(host.build_library(name, "fp32"))
    else:
        import numpy as np, torch
        # correctness first: every split of both libraries against the unsplit kernel of the default library (bitwise), ragged batch
        K = 16384 + 37
        x = np.random.default_rng(1).uniform(-1, 1, (K, 21)).astype(np.float32)
        d_in = torch.from_numpy(x).cuda()
        ref = {}
        for name in ["iiwa7"] + list(VARIANTS):
            h = host.GridHandle(name, precision="fp32"); n = h.n
            for alg, call in ((host.ALG_FD_DU, h.forward_dynamics_gradient_device), (host.ALG_ID_DU, h.inverse_dynamics_gradient_device)):
                for S in [1] + h.L.splits(alg):
                    for threads in (0, 64, 192):
                        d_out = torch.full((K, 2 * n * n), float("nan"), dtype=torch.float32, device='cuda')
                        h.set_split(alg, S)
                        call(d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, threads=threads)
                        torch.cuda.synchronize()
                        got = d_out.cpu().numpy()
                        if (name, S) == ("iiwa7", 1) and threads == 0:
                            ref[alg] = got
                        same = np.array_equal(got, ref[alg])
                        if not same:
                            print("MISMATCH", name, alg, S, threads, np.isnan(got).sum(), np.abs(got - ref[alg]).max())
            h.close()
        print("bitwise check of every split / block shape done", flush=True)
        for K in (64, 1024, 4096, 8192, 16384):
            for name in ["iiwa7"] + list(VARIANTS):
                h = host.GridHandle(name, precision="fp32"); n = h.n
                x = np.random.default_rng(0).uniform(-1, 1, (K, 3 * n)).astype(np.float32)
                d_in = torch.from_numpy(x).cuda(); d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device='cuda')
                row = []
                for alg, nm in ((host.ALG_FD_DU, "dFD"), (host.ALG_ID_DU, "dID")):
                    for S in (0,):
                        h.set_split(alg, S)
                        for threads, tag in ((0, "auto"),):
                            h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, threads=threads, reps=(300 if K <= 65536 else 20))
                            t = min(h.time_device(alg, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, threads=threads, reps=(300 if K <= 65536 else 20)) for _ in range(5)) * 1e3
                            row.append("%s auto(S=%d) %7.2f us" % (nm, h.get_split(alg, K), t))
                print("K=%-6d %-11s | %s" % (K, name, " | ".join(row)), flush=True)
                h.close()
