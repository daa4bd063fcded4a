"""BASELINE config 5's WHOLE batch on one GPU (Atlas-30, 1 048 576 configurations: 7.5 GB of output): the forward-dynamics gradient and the
inverse-dynamics gradient through the automatic choice (register-lean kernels); rows sampled over the whole range must equal, bit for
bit, what a small batch of the same configurations gives (64-bit row addressing, grid-stride over 16 384 tiles), and every value is
finite.  Run on the GPU box: `python tests/gpu_checks/atlas30_one_million.py`."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    import torch
    from gridcodegenerator_amd import host
    K = 1 << 20
    h = host.GridHandle("atlas30", precision="fp32"); n = h.n
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, (K, n)), rng.uniform(-1, 1, (K, 2 * n))], axis=1).astype(np.float32)
    d_in = torch.from_numpy(x).cuda()
    rows = np.unique(np.concatenate([np.arange(64), rng.integers(0, K, 256), np.arange(K - 64, K)]))
    d_small = torch.from_numpy(np.ascontiguousarray(x[rows])).cuda()
    for alg, name, call in ((host.ALG_FD_DU, "forward_dynamics_gradient", h.forward_dynamics_gradient_device),
                            (host.ALG_ID_DU, "inverse_dynamics_gradient", h.inverse_dynamics_gradient_device)):
        out = torch.empty((K, 2 * n * n), dtype=torch.float32, device="cuda")
        call(out.data_ptr(), d_in.data_ptr(), 3 * n, K); h.synchronize()
        ms = min(h.time_device(alg, out.data_ptr(), d_in.data_ptr(), 3 * n, K, reps=3) for _ in range(2))
        small = torch.empty((len(rows), 2 * n * n), dtype=torch.float32, device="cuda")
        h.set_wave(alg, 1)
        call(small.data_ptr(), d_small.data_ptr(), 3 * n, len(rows)); h.synchronize()
        h.set_wave(alg, 0)
        same = bool(torch.equal(out[torch.from_numpy(rows).cuda()], small))
        finite = bool(torch.isfinite(out).all().item())
        print("%s K = %d: kernel variant %d, %.2f ms per launch = %.1f M evals/s, sampled rows bit-identical to a %d-row batch: %s, all finite: %s"
              % (name, K, h.get_coop(alg, K), ms, K / ms / 1e3, len(rows), same, finite), flush=True)
        del out
    h.close()


if __name__ == "__main__":
    main()
