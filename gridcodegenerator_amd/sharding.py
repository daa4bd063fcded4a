"""Batch sharding + timing aggregation for multi-GPU runs.

The path shards trivially -- configurations are independent and the model is read-only -- so there is
NO collective on the data path.  ``torch.distributed`` (backend "nccl" = RCCL on the GPUs, "gloo" in the
CPU tests) is used only for the start/stop barrier and the max-over-ranks of the elapsed time.
"""
import os
import time


def shard_bounds(total, world, rank):
    """Contiguous slice [lo, hi) of a global batch for `rank`: [g*K/G, (g+1)*K/G)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (rank * total) // world, ((rank + 1) * total) // world


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend):
    """Returns the torch.distributed module (initialised) or None for a single process."""
    rank, _, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def visible_gpu_count(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process would see, counted WITHOUT touching the HIP / HSA runtime: the KFD topology in sysfs (a node with SIMDs is a
    GPU) cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (comma-separated indices or UUIDs; an empty
    value hides every device).  bench.py's launcher uses it so that the parent of the ranks never initialises the GPU
    (`torch.cuda.device_count()` may fall back to hipGetDeviceCount on builds without amdsmi).  Returns None when the topology
    cannot be read (no /sys/class/kfd: not a ROCm host)."""
    try:
        nodes = sorted(os.listdir(root), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    gpus = 0
    for nd in nodes:
        try:
            with open(os.path.join(root, nd, "properties")) as fh:
                props = dict(line.split(None, 1) for line in fh.read().splitlines() if " " in line)
        except OSError:
            continue            # (a node of another container's cgroup: not ours)
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            gpus += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in os.environ:
            listed = [x for x in os.environ[var].split(",") if x.strip() != ""]
            gpus = min(gpus, len(listed))
    return gpus


def timed_steps(step, steps, warmup, device_sync, dist=None, reduce_device=None, on_start=None, on_stop=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device_sync on both sides;
    returns the MAX over ranks of the elapsed seconds.  on_start / on_stop: called right before the first and right after the last
    timed step is enqueued (bench.py records device events on the launch stream there: the kernel time INSIDE the timed region)."""
    for _ in range(warmup):
        step()
    device_sync()
    if dist is not None:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    if on_start is not None:
        on_start()
    for _ in range(steps):
        step()
    if on_stop is not None:
        on_stop()
    device_sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_throughput(units_per_rank_per_step, world, steps, elapsed):
    """Whole-job throughput: the units all ranks processed divided by the max-over-ranks time."""
    return float(units_per_rank_per_step) * world * steps / elapsed


def run_shards_in_process(total, devices, make_shard, steps, warmup=0):
    """The same batch sharding WITHOUT a launcher: ONE process, one host thread per device, one handle and one stream per thread
    (the C ABI allows N handles in a process: include/grid_capi.h grid_init(device, ...)).  For a node where torch.distributed.run is
    not available; `bench.py --gpus N` keeps to one process per GPU.

    make_shard(index, device, lo, hi) -> (step, sync): called IN the shard's thread; `step()` enqueues one pass over configurations
    [lo, hi) on that device, `sync()` waits for the device's queued work.  Every thread does `warmup` untimed steps, meets the
    others at a barrier, runs `steps` timed steps and synchronises; the elapsed time is the max over the threads -- the thread
    analogue of timed_steps().  Returns dict(elapsed=..., per_shard=[(lo, hi, seconds)], value=configurations per second of the
    whole job).  No data crosses between shards; an exception in any shard is re-raised here."""
    import threading
    world = len(devices)
    if world < 1:
        raise ValueError("need at least one device")
    bounds = [shard_bounds(total, world, r) for r in range(world)]
    gate = threading.Barrier(world)
    times, errors = [None] * world, [None] * world

    def work(r):
        try:
            lo, hi = bounds[r]
            step, sync = make_shard(r, devices[r], lo, hi)
            for _ in range(warmup):
                step()
            sync()
            gate.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            sync()
            times[r] = time.perf_counter() - t0
        except BaseException as exc:       # noqa: BLE001  (reported to the caller's thread)
            errors[r] = exc
            gate.abort()

    threads = [threading.Thread(target=work, args=(r,), name="grid-shard-%d" % r) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    elapsed = max(times)
    return dict(elapsed=elapsed, per_shard=[(lo, hi, t) for (lo, hi), t in zip(bounds, times)],
                value=float(total) * steps / elapsed)
