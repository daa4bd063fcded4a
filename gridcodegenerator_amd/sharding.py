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


def timed_steps(step, steps, warmup, device_sync, dist=None, reduce_device=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device_sync on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step()
    device_sync()
    if dist is not None:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
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
