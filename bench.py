#!/usr/bin/env python3
"""Headline benchmark: forward-dynamics-gradient evaluations per second, iiwa-7, batch 16384 per GPU.

Contract (task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched
by ``python -m torch.distributed.run --nproc-per-node N ...`` with one rank per GPU.  One "step" is one
pass of the hot path -- ``forward_dynamics_gradient_kernel`` over one batch of synthetic (q, qd, u)
already resident in HBM -- launched through the C ABI on torch's current stream.  W untimed warm-up
steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize(); the time is the MAX over
ranks; rank 0 prints ONE JSON line.

Sharding: the batch dimension is embarrassingly parallel, every rank owns its own contiguous slice
(weak scaling: 16384 configurations per GPU), there is NO collective on the data path; torch.distributed
(RCCL) is used only for the barrier and the max-over-ranks of the elapsed time.

roofline: algorithmic bytes per launch = 4*(3n + 2n^2) * batch, divided by the
kernel's average launch duration measured with HIP events on the launch stream (grid_time_device).
cpu_baseline: the numpy oracle (oracle/rbd_oracle.py, a float64 port of the reference algorithm) timed on
this box's host cores (up to 16 worker processes, count reported), rank 0 at N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GRAVITY = 9.81


def make_inputs(n, K, seed):
    """q ~ U(-pi, pi), qd ~ U(-1, 1), u ~ U(-1, 1), float64 draw cast to fp32."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (K, n)).astype(np.float32)
    qd = rng.uniform(-1.0, 1.0, (K, n)).astype(np.float32)
    u = rng.uniform(-1.0, 1.0, (K, n)).astype(np.float32)
    return q, qd, u


def _cpu_worker(job):
    """One process of the CPU baseline: whole passes of the numpy oracle over its slice for about `seconds`."""
    robot_name, q64, qd64, u64, seconds = job
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)      # one BLAS/OpenMP thread per worker process
    except Exception:                              # pragma: no cover
        limiter = None
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    T = O.RobotTables(get_robot(robot_name))
    O.fd_grad(T, q64[:8], qd64[:8], u64[:8])
    passes = 0
    t0 = time.perf_counter()
    while passes == 0 or time.perf_counter() - t0 < seconds:
        O.fd_grad(T, q64, qd64, u64)
        passes += 1
    dt = time.perf_counter() - t0
    del limiter
    return passes * q64.shape[0], dt


def cpu_baseline(robot_name, q, qd, u, seconds, cores):
    """numpy float64 oracle on `cores` host cores: the batch is cut into `cores` contiguous slices, one forked worker process
    each (forked BEFORE anything touches the GPU), every worker repeats whole passes over its slice for about `seconds`.
    value = all evaluations done / the longest worker time.  cores = 1 runs in this process."""
    q64, qd64, u64 = (a.astype(np.float64) for a in (q, qd, u))
    K = q.shape[0]
    cores = max(1, min(int(cores), K))
    bounds = [(i * K) // cores for i in range(cores + 1)]
    jobs = [(robot_name, q64[a:b], qd64[a:b], u64[a:b], seconds) for a, b in zip(bounds[:-1], bounds[1:])]
    if cores == 1:
        results = [_cpu_worker(jobs[0])]
    else:
        import multiprocessing
        with multiprocessing.get_context("fork").Pool(cores) as pool:
            results = pool.map(_cpu_worker, jobs)
    evals = sum(r[0] for r in results)
    dt = max(r[1] for r in results)
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as fh:
            cpu = next(line.split(":", 1)[1].strip() for line in fh if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return dict(value=evals / dt, unit="evals/s", cores=cores, kind="port", cpu=cpu,
                host_cores_available=len(os.sched_getaffinity(0)),
                sample="oracle.fd_grad (numpy float64, batch-vectorised) over the same %d-configuration batch cut into %d slices, "
                       "one process per slice, whole passes for %.1f s: %d evaluations" % (K, cores, dt, evals))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--robot", default="iiwa7")
    ap.add_argument("--batch", type=int, default=16384, help="configurations per GPU")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall seconds of the cpu_baseline sample (whole passes)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the cpu_baseline (0 = all cores of this process, at most 16)")
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="seconds of untimed launches before the warm-up (clock ramp)")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--split", type=int, default=0, help="column-split factor of the gradient kernel: 0 auto, 1 never, S force")
    ap.add_argument("--all-kernels", action="store_true", help="also time the other four kernels (reported under 'kernels')")
    args = ap.parse_args()

    from gridcodegenerator_amd import host, sharding
    from gridcodegenerator_amd.robots import get_robot

    rank, local_rank, world = sharding.env_rank()
    # CPU baseline first: its worker processes are forked before this process initialises the GPU runtime
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline:
        n0 = get_robot(args.robot).get_num_joints()
        q0, qd0, u0 = make_inputs(n0, args.batch, 3 + rank)
        cores = args.cpu_cores if args.cpu_cores > 0 else min(16, len(os.sched_getaffinity(0)))
        cpu_line = cpu_baseline(args.robot, q0, qd0, u0, args.cpu_seconds, cores)

    import torch
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal on a box with fewer GPUs than ranks (GRID_BENCH_REHEARSAL=1): every rank uses device 0 and the barrier /
    # max-reduction run over gloo -- exercises the multi-process path (build lock, per-rank handles, aggregation) only
    rehearsal = os.environ.get("GRID_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = sharding.init_distributed("gloo" if rehearsal else "nccl")

    host.build_library(args.robot, args.precision)
    h = host.GridHandle(args.robot, device=local_rank, precision=args.precision)
    n, K = h.n, args.batch
    # every rank owns an independent slice of the global batch (seed differs per rank)
    q, qd, u = make_inputs(n, K, 3 + rank)
    x = np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty((K, 2 * n * n), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    h.set_split(host.ALG_FD_DU, args.split)
    split_used = h.get_split(host.ALG_FD_DU, K)

    def step():
        h.forward_dynamics_gradient_device(d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, gravity=GRAVITY,
                                           blocks=args.blocks, threads=args.threads, stream=stream)

    # Bring the GPU out of its idle power state first (an MI355X box idles in a low-power state and the default run is only
    # ~3 ms of kernels): launch the same step for --prewarm-s seconds of wall clock.  Not counted as warm-up or steps.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_s:
        for _ in range(50):
            step()
        torch.cuda.synchronize()

    elapsed = sharding.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dist, reduce_device="cpu" if rehearsal else "cuda")

    # in-stream kernel duration (HIP events recorded on the launch stream by the C ABI)
    kern_ms = h.time_device(host.ALG_FD_DU, d_out.data_ptr(), d_in.data_ptr(), 3 * n, K, gravity=GRAVITY,
                            blocks=args.blocks, threads=args.threads, stream=stream, reps=max(20, min(args.steps, 200)))   # average over reps
    finite = bool(torch.isfinite(d_out).all().item())

    if rank == 0:
        value = sharding.aggregate_throughput(K, world, args.steps, elapsed)
        traffic = None      # PMC counters cannot be read from inside the timed process; use the committed rocprofv3 pass
        try:
            with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as fh:
                traffic = json.load(fh).get("%s:%d:forward_dynamics_gradient" % (args.robot, K), {}).get("bytes")
        except OSError:
            pass
        alg_bytes = host.algorithmic_bytes(host.ALG_FD_DU, n) * K
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        attrs = h.L.kernel_attributes(host.ALG_FD_DU)
        out = {
            "metric": "FD-gradient evals/sec (iiwa-7, batch=16k) + achieved HBM GB/s vs peak" if args.robot == "iiwa7" and K == 16384
                      else "FD-gradient evals/sec (%s, batch=%d)" % (args.robot, K),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": h.L.compute_dtype, "data": "synthetic",
            "config": {"workload": "%s forward_dynamics_gradient_kernel, batch %d per GPU, fp32 I/O, device-resident (reference _compute_only)"
                                   % (args.robot, K),
                       "robot": args.robot, "num_joints": n, "batch_per_gpu": K, "global_batch": K * world,
                       "parallelism": "batch-sharded x%d, independent streams, no collective on the data path" % world,
                       "launch": {"blocks": args.blocks or "suggested", "threads": args.threads or h.L.constants["SUGGESTED_THREADS"],
                                  "column_split": split_used},
                       "kernel": {"vgprs": attrs["numRegs"], "scratch_bytes_per_lane": attrs["scratch_bytes_per_lane"]},
                       "outputs_finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, profiles/)", "kernel": "forward_dynamics_gradient_kernel", "kernel_avg_us": 1e3 * kern_ms,
                         "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "algorithmic_bytes_per_eval": host.algorithmic_bytes(host.ALG_FD_DU, n),
                         "kernel_evals_per_s": K / (kern_ms * 1e-3)},
        }
        if args.all_kernels:
            kern = {}
            bufs = {a: torch.empty((K, host.output_size(a, n)), dtype=torch.float32, device="cuda") for a in range(5)}
            for a in range(5):
                h.time_device(a, bufs[a].data_ptr(), d_in.data_ptr(), 3 * n, K, gravity=GRAVITY, stream=stream, reps=200)   # ramp
                ms = h.time_device(a, bufs[a].data_ptr(), d_in.data_ptr(), 3 * n, K, gravity=GRAVITY, stream=stream, reps=200)
                by = host.algorithmic_bytes(a, n) * K
                kern[host.ALG_NAMES[a]] = {"avg_us": 1e3 * ms, "evals_per_s": K / (ms * 1e-3), "alg_GBps": by / (ms * 1e-3) / 1e9}
            out["kernels"] = kern
        if cpu_line is not None:
            out["cpu_baseline"] = cpu_line
        print(json.dumps(out), flush=True)
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
