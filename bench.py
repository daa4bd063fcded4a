#!/usr/bin/env python3
"""Headline benchmark: forward-dynamics-gradient evaluations per second, iiwa-7, batch 16384 per GPU.

Contract (task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched
by ``python -m torch.distributed.run --nproc-per-node N ...`` with one rank per GPU.  One "step" is one
pass of the hot path -- ``forward_dynamics_gradient_kernel`` over one batch of synthetic (q, qd, u)
already resident in HBM -- launched through the C ABI on torch's current stream.  W untimed warm-up
steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize(); the time is the MAX over
ranks; rank 0 prints ONE JSON line.

The line's top-level fields are the headline workload (BASELINE.json: iiwa-7, batch 16384 per GPU).  north_star's target
also names Atlas-30, so the same run times it too, with the same bracket, and reports it under ``"secondary"``:
Atlas-30 batch 16384 per GPU always, and for N > 1 additionally BASELINE config 5's shard (Atlas-30, 131072 per GPU).

Sharding: the batch dimension is embarrassingly parallel, every rank owns its own contiguous slice
(weak scaling), there is NO collective on the data path; torch.distributed (RCCL) is used only for the
barrier and the max-over-ranks of the elapsed time.

roofline: algorithmic bytes per launch = 4*(3n + 2n^2) * batch, divided by the
kernel's average launch duration.  Three clocks see a launch in this run -- the host's wall clock per step of the timed region, HIP
events recorded on the launch stream around a repetition of exactly that region (same K steps; the events' host cost stays out of the
timed region), and HIP events around a further set of back-to-back launches (grid_time_device) --
and ``roofline.achieved`` / ``frac`` are priced on the SLOWEST of them; ``roofline.clock`` names it, ``clocks_us`` lists all three.
``secondary`` holds the other single-GPU configurations of BASELINE.json (C2, C3's forward dynamics, C4) and north_star's Atlas-30
batch-16k target, each with its own ``roofline``.  ``kernel`` is the
kernel the C ABI actually dispatched (e.g. ``..._kernel_split4``) with that kernel's registers.  ``traffic`` comes from
the committed rocprofv3 PMC passes (profiles/pmc_traffic.json) and is printed only when that entry was measured on the
same generated header (sha recorded with the entry) -- otherwise null.
cpu_baseline: the numpy oracle (oracle/rbd_oracle.py, a float64 port of the reference algorithm) timed on
this box's host cores (count reported, see host_cores()), rank 0 at N=1 only, on a bounded sample.
"""
import argparse
import hashlib
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
HEADLINE_METRIC = "FD-gradient evals/sec (iiwa-7, batch=16k) + achieved HBM GB/s vs peak"


def make_inputs(n, K, seed):
    """q ~ U(-pi, pi), qd ~ U(-1, 1), u ~ U(-1, 1), float64 draw cast to fp32."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (K, n)).astype(np.float32)
    qd = rng.uniform(-1.0, 1.0, (K, n)).astype(np.float32)
    u = rng.uniform(-1.0, 1.0, (K, n)).astype(np.float32)
    return q, qd, u


def host_cores():
    """Cores this process may really use: the scheduler affinity mask, cut down to the cgroup CPU quota when the container
    has one (a one-GPU box exposes all 256 hardware threads in the mask but grants a share of them)."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    return (min(affinity, quota) if quota else affinity), affinity, quota


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
    cores = max(1, min(int(cores), K // 8))
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
    usable, affinity, quota = host_cores()
    # the same leg also yields the checker values for the parity figures of the line (first 256 configurations, float64)
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    ns = min(256, K)
    T = O.RobotTables(get_robot(robot_name))
    c_ref = O.rnea(T, q64[:ns], qd64[:ns])[0]
    qdd_ref = O.forward_dynamics(T, q64[:ns], qd64[:ns], u64[:ns])
    globals()["_PARITY_REF"] = dict(n=ns, c=c_ref, qdd=qdd_ref)
    return dict(value=evals / dt, unit="evals/s", cores=cores, kind="port", cpu=cpu,
                host_cores_usable=usable, host_cores_affinity=affinity, cgroup_cpu_quota=quota,
                sample="oracle.fd_grad (numpy float64, batch-vectorised) over the same %d-configuration batch cut into %d slices, "
                       "one process per slice, whole passes for %.1f s: %d evaluations" % (K, cores, dt, evals))


def header_sha(robot, precision):
    from gridcodegenerator_amd import host
    try:
        with open(host.library_paths(robot, precision)["header"], "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def committed_traffic(robot, K, kernel, sha):
    """HBM bytes per launch of `kernel` from the committed PMC passes -- only if they were taken on this very header."""
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as fh:
            e = json.load(fh).get("%s:%d:%s" % (robot, K, kernel))
    except (OSError, ValueError):
        return None, None
    if not e or e.get("header_sha") != sha:
        return None, None
    return e.get("bytes"), e.get("round")


def launch_ranks(n_gpus, argv):
    """Run `bench.py --gpus N ...` as N ranks (python -m torch.distributed.run, one rank per GPU) from a process that has not
    initialised the GPU; returns the exit code.  Fewer than N devices: an error, unless GRID_BENCH_REHEARSAL=1 asks for the
    rehearsal (every rank on device 0, gloo).  GRID_BENCH_DRY_RUN=1 prints the command instead of running it (tests)."""
    import socket
    import subprocess
    from gridcodegenerator_amd import sharding
    # counted from sysfs (KFD topology + *_VISIBLE_DEVICES): this process must never initialise HIP/HSA -- the ranks are its children
    # (torch.cuda.device_count() can fall back to hipGetDeviceCount, which does).  No readable topology: let the ranks find out.
    have = sharding.visible_gpu_count()
    if have is None:        # no KFD topology in sysfs: without /dev/kfd there is no ROCm GPU; with it, one per render node
        import glob
        have = len(glob.glob("/dev/dri/renderD*")) if os.path.exists("/dev/kfd") else 0
    rehearsal = os.environ.get("GRID_BENCH_REHEARSAL") == "1"
    if have < n_gpus and not rehearsal:
        print("bench.py: --gpus %d requested but %d GPU(s) are visible; refusing to report a %d-GPU number from fewer devices "
              "(GRID_BENCH_REHEARSAL=1 runs the multi-rank path on device 0 instead)" % (n_gpus, have, n_gpus), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if os.environ.get("GRID_BENCH_DRY_RUN") == "1":
        print(" ".join(cmd))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)            # the ranks' stdout (rank 0's JSON line) and stderr pass through


class Workload:
    """One robot / batch on this rank's GPU: device-resident inputs and outputs, one handle, the launch closure."""

    def __init__(self, torch, host, robot, K, precision, device, seed, blocks=0, threads=0, split=0, coop=0, streams=1, wave=0, alg=None):
        host.build_library(robot, precision)
        self.host, self.torch, self.robot, self.K, self.precision = host, torch, robot, K, precision
        self.alg = host.ALG_FD_DU if alg is None else alg
        self.h = host.GridHandle(robot, device=device, precision=precision)
        self.n = n = self.h.n
        q, qd, u = make_inputs(n, K, seed)
        self.d_in = torch.from_numpy(np.ascontiguousarray(np.concatenate([q, qd, u], axis=1))).cuda()
        self.d_out = torch.empty((K, host.output_size(self.alg, n)), dtype=torch.float32, device="cuda")
        # torch's current stream (0 = the default stream: GridHandle passes it on as hipStreamLegacy; a NULL pointer would mean the
        # handle's own non-blocking stream to the C ABI, which is not ordered with torch's work)
        self.stream = torch.cuda.current_stream().cuda_stream
        # streams > 1: consecutive steps (independent batches) go round-robin over streams created here, each with its own output
        # buffer, so that the dispatch / end-of-kernel gap of one launch overlaps with the next one's execution.  None of them is
        # the default stream: work on the default stream waits for every blocking stream and would serialise the round-robin.
        self.n_streams = max(1, int(streams))
        self.extra_streams = [torch.cuda.Stream() for _ in range(self.n_streams if self.n_streams > 1 else 0)]
        self.stream_ptrs = [st.cuda_stream for st in self.extra_streams] or [self.stream]
        self.outs = [self.d_out] + [torch.empty_like(self.d_out) for _ in range(self.n_streams - 1)]
        self.step_no = 0
        self.blocks, self.threads = blocks, threads
        self.h.set_split(self.alg, split)
        if coop:
            self.h.set_coop(self.alg, coop)
        if wave:
            self.h.set_wave(self.alg, wave)
        # precedence in the C ABI: wave-per-configuration kernel (small batches), tile-cooperative kernel, column split; an explicit
        # launch shape keeps the lane-per-configuration kernels (include/grid_capi.h)
        self.wave_used = self.h.get_wave(self.alg, K) and not (blocks or threads)
        self.coop_used = (not self.wave_used) and self.h.get_coop(self.alg, K)
        self.split_used = 1 if (self.coop_used or self.wave_used) else self.h.get_split(self.alg, K)

    def launch(self, d_out_ptr, stream):
        """One pass of the algorithm over the batch through the C ABI (device pointers: the reference's _compute_only mode)."""
        host, h, n = self.host, self.h, self.n
        kw = dict(blocks=self.blocks, threads=self.threads, stream=stream)
        if self.alg == host.ALG_FD_DU:
            h.forward_dynamics_gradient_device(d_out_ptr, self.d_in.data_ptr(), 3 * n, self.K, gravity=GRAVITY, **kw)
        elif self.alg == host.ALG_ID_DU:
            h.inverse_dynamics_gradient_device(d_out_ptr, self.d_in.data_ptr(), 3 * n, self.K, gravity=GRAVITY, **kw)
        elif self.alg == host.ALG_FD:
            h.forward_dynamics_device(d_out_ptr, self.d_in.data_ptr(), 3 * n, self.K, gravity=GRAVITY, **kw)
        elif self.alg == host.ALG_MINV:
            h.direct_minv_device(d_out_ptr, self.d_in.data_ptr(), 3 * n, self.K, **kw)
        else:
            h.inverse_dynamics_device(d_out_ptr, self.d_in.data_ptr(), 3 * n, self.K, gravity=GRAVITY, **kw)

    def step(self):
        i = self.step_no % self.n_streams
        self.step_no += 1
        self.launch(self.outs[i].data_ptr(), self.stream_ptrs[i])

    def prewarm(self, seconds):
        # Bring the GPU out of its idle power state (the default run is only a few ms of kernels).  Not warm-up, not steps.
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(50 if self.K <= 65536 else 5):
                self.step()
            self.torch.cuda.synchronize()

    def measure(self, sharding, dist, steps, warmup, world, reduce_device):
        host, torch = self.host, self.torch
        # HIP events on the launch stream around the timed region itself (torch events are recorded on torch's current stream, which
        # IS the launch stream here: main() makes a stream of its own current and Workload.stream is that stream)
        # the timed region proper: barrier + synchronize, exactly K steps, synchronize + barrier -- nothing else inside (two event
        # records cost the host ~10 us, 5 % of a 20-step region)
        elapsed = sharding.timed_steps(self.step, steps, warmup, torch.cuda.synchronize, dist, reduce_device=reduce_device)
        # ... and the same K steps once more with HIP events recorded on the launch stream around them: the kernel time of the region
        ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] if self.n_streams == 1 else None
        region_ms = None
        if ev:
            ev[0].record(); ev[1].record(); torch.cuda.synchronize()       # (a torch event is created by its first record())
            sharding.timed_steps(self.step, steps, 0, torch.cuda.synchronize, None, on_start=lambda: ev[0].record(), on_stop=lambda: ev[1].record())
            region_ms = ev[0].elapsed_time(ev[1]) / steps
        reps = max(20, min(steps, 200)) if self.K <= 65536 else max(5, min(steps, 20))
        b2b_ms = self.h.time_device(self.alg, self.d_out.data_ptr(), self.d_in.data_ptr(), 3 * self.n, self.K, gravity=GRAVITY,
                                    blocks=self.blocks, threads=self.threads, stream=self.stream, reps=reps)
        finite = bool(torch.isfinite(self.d_out).all().item())
        n, K = self.n, self.K
        kernel = host.ALG_NAMES[self.alg] + "_kernel" + ("_wave" if self.wave_used else "_coop8" if self.coop_used == 2 else "_coop" if self.coop_used else
                                                          ("_split%d" % self.split_used if self.split_used > 1 else ""))
        attrs = self.h.L.kernel_attributes(self.alg, split=self.split_used, coop=self.coop_used, wave=self.wave_used)
        sha = header_sha(self.robot, self.precision)
        traffic, traffic_round = committed_traffic(self.robot, K, kernel, sha)
        alg_bytes = host.algorithmic_bytes(self.alg, n) * K
        # roofline.achieved is priced on the SLOWEST of the clocks this run has for one launch, and says which: HIP events around the
        # timed region, HIP events around `reps` more back-to-back launches (grid_time_device), the host's wall clock per step
        clocks = {"hip_events_back_to_back": b2b_ms, "wall_per_step": 1e3 * elapsed / steps}
        if region_ms is not None:
            clocks["hip_events_timed_region"] = region_ms
        clock = max(clocks, key=lambda c: clocks[c]) if self.n_streams == 1 else "hip_events_back_to_back"
        kern_ms = clocks[clock]
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        return {
            "value": sharding.aggregate_throughput(K, world, steps, elapsed), "unit": "evals/s", "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps, "dtype": self.h.L.compute_dtype,
            "config": {"workload": "%s %s_kernel, batch %d per GPU, fp32 I/O, device-resident (reference _compute_only)"
                                   % (self.robot, host.ALG_NAMES[self.alg], K),
                       "robot": self.robot, "num_joints": n, "batch_per_gpu": K, "global_batch": K * world,
                       "parallelism": "batch-sharded x%d, independent streams, no collective on the data path%s"
                                      % (world, "" if self.n_streams == 1 else "; steps round-robin over %d streams per GPU" % self.n_streams),
                       "launch": {"blocks": self.blocks or "suggested", "threads": self.threads or self.h.L.constants["SUGGESTED_THREADS"],
                                  "column_split": self.split_used, "tile_cooperative": {0: False, 1: "4 waves per tile", 2: "8 waves per tile (register-lean)"}[int(self.coop_used)],
                                  "wave_per_configuration": bool(self.wave_used)},
                       "kernel": {"name": kernel, "vgprs": attrs["numRegs"], "scratch_bytes_per_lane": attrs["scratch_bytes_per_lane"],
                                  "header_sha": sha},
                       "outputs_finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE passes of %s on this header; null = not measured on this header)"
                                         % (traffic_round or "profiles/"),
                         "kernel": kernel, "kernel_avg_us": 1e3 * kern_ms, "clock": clock,
                         "clocks_us": {c: 1e3 * v for c, v in clocks.items()},
                         "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "algorithmic_bytes_per_eval": host.algorithmic_bytes(self.alg, n),
                         "kernel_evals_per_s": K / (kern_ms * 1e-3)},
        }

    def parity_sample(self, ref):
        """Norm-wise and worst element-wise (entries >= 1e-3 of the scale) error of the torques c and accelerations qdd -- what
        north_star's 1e-6 bar names -- of the first ref["n"] configurations of this batch against the float64 checker values from
        the cpu_baseline leg (report both figures and say which bar is met)."""
        host, torch, n = self.host, self.torch, self.n
        ns = ref["n"]
        out = {}
        for (name, alg, width) in (("c", host.ALG_ID, n), ("qdd", host.ALG_FD, n)):
            buf = torch.empty((ns, width), dtype=torch.float32, device="cuda")
            if alg == host.ALG_ID:
                self.h.inverse_dynamics_device(buf.data_ptr(), self.d_in.data_ptr(), 3 * n, ns, gravity=GRAVITY, stream=self.stream)
            else:
                self.h.forward_dynamics_device(buf.data_ptr(), self.d_in.data_ptr(), 3 * n, ns, gravity=GRAVITY, stream=self.stream)
            torch.cuda.synchronize()
            got = buf.cpu().numpy().astype(np.float64)
            err = np.abs(got - ref[name])
            scale = np.abs(ref[name]).max()
            big = np.abs(ref[name]) >= 1e-3 * scale
            out[name] = {"normwise": float(err.max() / scale), "elementwise_above_1e-3_of_scale": float((err[big] / np.abs(ref[name][big])).max())}
        out["bar"] = "north_star 1e-6 relative is met NORM-WISE (max|err| / max|ref|); element-wise the entries above 1e-3 of the scale are within the second figure"
        out["configurations"] = ns
        return out

    def all_kernels(self):
        host, torch, n, K = self.host, self.torch, self.n, self.K
        kern = {}
        bufs = {a: torch.empty((K, host.output_size(a, n)), dtype=torch.float32, device="cuda") for a in range(5)}
        for a in range(5):
            self.h.time_device(a, bufs[a].data_ptr(), self.d_in.data_ptr(), 3 * n, K, gravity=GRAVITY, stream=self.stream, reps=200)   # ramp
            ms = self.h.time_device(a, bufs[a].data_ptr(), self.d_in.data_ptr(), 3 * n, K, gravity=GRAVITY, stream=self.stream, reps=200)
            by = host.algorithmic_bytes(a, n) * K
            kern[host.ALG_NAMES[a]] = {"avg_us": 1e3 * ms, "evals_per_s": K / (ms * 1e-3), "alg_GBps": by / (ms * 1e-3) / 1e9}
        return kern

    def close(self):
        self.h.close()
        del self.d_in, self.d_out, self.outs


def main():
    from gridcodegenerator_amd import host, sharding
    from gridcodegenerator_amd.robots import get_robot
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--robot", default="iiwa7")
    ap.add_argument("--batch", type=int, default=16384, help="configurations per GPU")
    ap.add_argument("--precision", default=host.DEFAULT_PRECISION, choices=list(host.VERIFIED_PRECISIONS),
                    help="arithmetic of the kernels (fp64 is not verified on the GPU and not offered)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Atlas-30 workloads reported under 'secondary'")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall seconds of the cpu_baseline sample (whole passes)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the cpu_baseline (0 = every core this process may use)")
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="seconds of untimed launches before the warm-up (clock ramp)")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--split", type=int, default=0, help="column-split factor of the gradient kernel: 0 auto, 1 never, S force")
    ap.add_argument("--coop", type=int, default=0, help="tile-cooperative gradient kernel: 0 auto, 1 never, 2 always (4 waves), 3 always the register-lean 8-wave variant")
    ap.add_argument("--all-kernels", action="store_true", help="also time the other four kernels (reported under 'kernels')")
    args = ap.parse_args()

    rank, local_rank, world = sharding.env_rank()
    if args.gpus > 1 and world == 1:
        # `--gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU yet (and never will): the N
        # ranks are children started through torch.distributed.run, rank 0's JSON line is relayed, the exit code is theirs.  A
        # request that cannot be honoured fails loudly -- it is never answered with an `n_gpus: 1` line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if world > 1 and world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # CPU baseline first: its worker processes are forked before this process initialises the GPU runtime
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline:
        n0 = get_robot(args.robot).get_num_joints()
        q0, qd0, u0 = make_inputs(n0, args.batch, 3 + rank)
        cores = args.cpu_cores if args.cpu_cores > 0 else host_cores()[0]
        cpu_line = cpu_baseline(args.robot, q0, qd0, u0, args.cpu_seconds, cores)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal on a box with fewer GPUs than ranks (GRID_BENCH_REHEARSAL=1): every rank uses device 0 and the barrier /
    # max-reduction run over gloo -- exercises the multi-process path (build lock, per-rank handles, aggregation) only
    rehearsal = os.environ.get("GRID_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # a stream of its own as torch's current stream: torch's work and the launches (Workload.stream) are ordered on it, and the
    # launches avoid the default stream's implicit synchronisation with every other blocking stream
    torch.cuda.set_stream(torch.cuda.Stream())
    dist = sharding.init_distributed("gloo" if rehearsal else "nccl")
    reduce_device = "cpu" if rehearsal else "cuda"

    # ---- headline workload: every rank owns an independent slice of the global batch (seed differs per rank)
    w = Workload(torch, host, args.robot, args.batch, args.precision, local_rank, 3 + rank, args.blocks, args.threads, args.split, args.coop)
    w.prewarm(args.prewarm_s)
    main_line = w.measure(sharding, dist, args.steps, args.warmup, world, reduce_device)
    if rank == 0 and "_PARITY_REF" in globals():
        main_line["config"]["parity_sample"] = w.parity_sample(globals()["_PARITY_REF"])
    kernels = w.all_kernels() if (args.all_kernels and rank == 0) else None
    w.close()

    # ---- secondary workloads (north_star: "... iiwa-7 and Atlas-30 at batch 16k on 1 GPU and batch-sharded at 2/4/8 GPUs";
    #      BASELINE config 5: Atlas-30, 1,048,576 configurations over 8 GPUs = 131072 per GPU)
    secondary = {}
    if not args.no_secondary and args.robot == "iiwa7" and args.batch == 16384:
        # the headline workload once more with consecutive steps on two streams: the kernels themselves still serialise (one wave
        # per SIMD), but the ~2.6 us between dependent launches of a single stream overlap.  Reported beside `value`, never as it.
        w1 = Workload(torch, host, args.robot, args.batch, args.precision, local_rank, 3 + rank, args.blocks, args.threads, args.split, args.coop, streams=2)
        w1.prewarm(min(args.prewarm_s, 0.2))
        secondary["iiwa7_batch16384_two_streams"] = w1.measure(sharding, dist, args.steps, args.warmup, world, reduce_device)
        w1.close()
        # ... Atlas-30 at batch 16k, and its small-batch path (64 configurations, dispatched to the
        # wave-per-configuration kernel -- `ms_per_step` is the latency of one dependent launch)
        FD_DU, ID_DU, ID, FD = host.ALG_FD_DU, host.ALG_ID_DU, host.ALG_ID, host.ALG_FD
        plan = [("atlas30_batch16384", "atlas30", 16384, min(args.steps, 50), min(args.warmup, 5), FD_DU),
                ("atlas30_batch64_small_batch_path", "atlas30", 64, min(args.steps, 50), min(args.warmup, 5), FD_DU)]
        if world == 1:
            # the other single-GPU configurations of BASELINE.json, each with its own roofline: C2 (iiwa-7 RNEA + its gradient, batch
            # 1024), C3's forward dynamics (its gradient is the headline), C4 (Atlas-30 ID and FD gradients, batch 65536)
            plan += [("C2_iiwa7_batch1024_inverse_dynamics", "iiwa7", 1024, min(args.steps, 50), min(args.warmup, 5), ID),
                     ("C2_iiwa7_batch1024_inverse_dynamics_gradient", "iiwa7", 1024, min(args.steps, 50), min(args.warmup, 5), ID_DU),
                     ("C3_iiwa7_batch16384_forward_dynamics", "iiwa7", 16384, min(args.steps, 50), min(args.warmup, 5), FD),
                     ("C4_atlas30_batch65536_forward_dynamics_gradient", "atlas30", 65536, min(args.steps, 25), min(args.warmup, 3), FD_DU),
                     ("C4_atlas30_batch65536_inverse_dynamics_gradient", "atlas30", 65536, min(args.steps, 25), min(args.warmup, 3), ID_DU),
                     # C5 (Atlas-30, 1 048 576 configurations over 8 GPUs) as ONE GPU sees it: its shard of 131 072
                     ("C5_atlas30_batch131072_one_shard_of_8", "atlas30", 131072, min(args.steps, 10), min(args.warmup, 2), FD_DU)]
        if world > 1:
            plan.append(("atlas30_batch131072_per_gpu", "atlas30", 131072, min(args.steps, 10), min(args.warmup, 2), FD_DU))
        for (key, robot, K, steps, warmup, alg) in plan:
            w2 = Workload(torch, host, robot, K, args.precision, local_rank, 5 + rank, alg=alg)
            w2.prewarm(min(args.prewarm_s, 0.2))
            line = w2.measure(sharding, dist, steps, warmup, world, reduce_device)
            w2.close()
            secondary[key] = line

    if rank == 0:
        K = args.batch
        out = {"metric": HEADLINE_METRIC if (args.robot == "iiwa7" and K == 16384) else "FD-gradient evals/sec (%s, batch=%d)" % (args.robot, K),
               "value": main_line["value"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": main_line["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": main_line["dtype"], "data": "synthetic", "config": main_line["config"], "roofline": main_line["roofline"]}
        if secondary:
            out["secondary"] = secondary
        if kernels:
            out["kernels"] = kernels
        if cpu_line is not None:
            out["cpu_baseline"] = cpu_line
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
