"""N > 1 host logic on the CPU with the gloo backend, world_size 2: shard assignment, barrier-bracketed
timing with max-over-ranks, whole-job throughput -- exactly the functions bench.py uses on the GPUs
(there with backend "nccl" = RCCL).  There is no data-path collective to test: shards never exchange data."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from gridcodegenerator_amd import sharding

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_the_batch():
    for total in (1, 7, 16384, 1048576, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_bounds(1048576, 8, 3) == (393216, 524288)   # BASELINE.json configs[4]: 8 shards of 131072


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %(repo)r)
    import numpy as np
    from gridcodegenerator_amd import sharding
    from gridcodegenerator_amd.robots import get_robot
    from oracle import rbd_oracle as O
    rank, local_rank, world = sharding.env_rank()
    dist = sharding.init_distributed("gloo")
    assert dist is not None and dist.get_world_size() == 2
    total = 37
    lo, hi = sharding.shard_bounds(total, world, rank)
    rng = np.random.default_rng(5)
    q = rng.uniform(-3, 3, (total, 7)); qd = rng.uniform(-1, 1, (total, 7)); u = rng.uniform(-1, 1, (total, 7))
    T = O.RobotTables(get_robot("iiwa7"))
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one
    elapsed = sharding.timed_steps(step, steps=5, warmup=2, device_sync=lambda: None, dist=dist)
    out = O.fd_grad(T, q[lo:hi], qd[lo:hi], u[lo:hi])    # this rank's shard only; nothing is exchanged
    np.save(os.path.join(%(out)r, "shard%%d.npy" %% rank), out)
    with open(os.path.join(%(out)r, "rank%%d.json" %% rank), "w") as fh:
        json.dump(dict(rank=rank, lo=lo, hi=hi, elapsed=elapsed, calls=len(calls),
                       value=sharding.aggregate_throughput(hi - lo, world, 5, elapsed)), fh)
    dist.destroy_process_group()
""")


def test_two_rank_gloo_run(tmp_path, tables):
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(repo=REPO, out=str(tmp_path)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    info = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert (info[0]["lo"], info[0]["hi"], info[1]["lo"], info[1]["hi"]) == (0, 18, 18, 37)
    assert info[0]["calls"] == info[1]["calls"] == 7                      # W=2 untimed + exactly K=5 timed steps
    assert abs(info[0]["elapsed"] - info[1]["elapsed"]) < 1e-9            # both ranks hold the MAX over ranks
    assert info[0]["elapsed"] >= 5 * 0.04 * 0.9                           # ... which is the slow rank's time
    # concatenated shards == the un-sharded computation (no cross-shard coupling)
    from oracle import rbd_oracle as O
    rng = np.random.default_rng(5)
    q = rng.uniform(-3, 3, (37, 7)); qd = rng.uniform(-1, 1, (37, 7)); u = rng.uniform(-1, 1, (37, 7))
    full = O.fd_grad(tables("iiwa7"), q, qd, u)
    got = np.concatenate([np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")])
    assert np.array_equal(got, full)


def _bench(args, **env):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_n_without_a_launcher_cannot_report_one_gpu():
    """`bench.py --gpus N` run directly (no WORLD_SIZE) becomes the launcher of N ranks -- before anything touches the GPU -- or fails
    loudly; it never prints an `n_gpus: 1` line for an N-GPU request.  No GPU is visible here, so: (a) the plain request is refused
    with a non-zero exit code and no JSON line; (b) the rehearsal request builds the torch.distributed.run command of N ranks
    (dry run: the command is printed, not executed)."""
    run = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert run.returncode != 0 and "--gpus 2 requested" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
    run = _bench(["--gpus", "4", "--steps", "3", "--warmup", "1"], GRID_BENCH_REHEARSAL="1", GRID_BENCH_DRY_RUN="1")
    assert run.returncode == 0, run.stderr
    cmd = run.stdout.split()
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    run = _bench(["--gpus", "2", "--steps", "3"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    assert run.returncode != 0 and "does not match WORLD_SIZE" in (run.stderr + run.stdout)


def test_visible_gpu_count_reads_sysfs_and_visibility_masks(tmp_path, monkeypatch):
    """The launcher counts GPUs from the KFD topology (nodes with SIMDs; CPU nodes have simd_count 0) cut down by the *_VISIBLE_DEVICES
    variables -- no HIP call.  Fake topology: one CPU node, three GPU nodes, one unreadable node."""
    root = tmp_path / "nodes"
    for i, simd in enumerate([0, 1024, 1024, 1024]):
        d = root / str(i); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    (root / "4").mkdir()                                   # no properties file (another cgroup's device)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert sharding.visible_gpu_count(str(root)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert sharding.visible_gpu_count(str(root)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert sharding.visible_gpu_count(str(root)) == 1
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert sharding.visible_gpu_count(str(root)) == 0
    assert sharding.visible_gpu_count(str(tmp_path / "absent")) is None


def test_the_launcher_never_touches_torch_cuda():
    """bench.launch_ranks() runs in the process that becomes the parent of the ranks: it must not initialise (or even import) the GPU
    side of torch -- checked in a fresh interpreter with the dry-run switch."""
    code = ("import os, sys; sys.argv = ['bench.py']; sys.path.insert(0, %r); os.environ['GRID_BENCH_REHEARSAL'] = '1'; "
            "os.environ['GRID_BENCH_DRY_RUN'] = '1'; import bench; rc = bench.launch_ranks(2, ['--gpus', '2']); "
            "assert rc == 0; assert 'torch' not in sys.modules, 'the launcher imported torch'; print('clean')" % REPO)
    run = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip().endswith("clean"), run.stderr


def test_in_process_shards_on_host_threads():
    """run_shards_in_process: one host thread per device, contiguous slices, the elapsed time is the slowest thread's, every shard's
    steps are counted, and a failing shard's exception reaches the caller (CPU stand-ins for the per-device handles)."""
    import time
    log = []

    def make_shard(index, device, lo, hi):
        calls = []
        log.append((index, device, lo, hi, calls))
        return (lambda: (calls.append(1), time.sleep(0.01 * (index + 1)))), (lambda: None)
    res = sharding.run_shards_in_process(1000, ["gpu0", "gpu1", "gpu2"], make_shard, steps=4, warmup=2)
    log.sort()
    assert [(i, d, lo, hi) for (i, d, lo, hi, _) in log] == [(0, "gpu0", 0, 333), (1, "gpu1", 333, 666), (2, "gpu2", 666, 1000)]
    assert all(len(calls) == 6 for (*_, calls) in log)
    assert res["elapsed"] >= 4 * 0.03 * 0.9 and res["elapsed"] == max(t for (_, _, t) in res["per_shard"])
    assert abs(res["value"] - 1000 * 4 / res["elapsed"]) < 1e-6

    def bad(index, device, lo, hi):
        if index == 1:
            raise RuntimeError("shard 1 failed")
        return (lambda: None), (lambda: None)
    import pytest
    with pytest.raises(RuntimeError, match="shard 1 failed"):
        sharding.run_shards_in_process(10, [0, 1], bad, steps=1)
