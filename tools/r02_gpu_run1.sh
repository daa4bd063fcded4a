#!/bin/bash
# Round-2 GPU session 1: environment, GPU tests, precision report (fp32 vs mixed), kernel-variant sweep, bench.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02a; mkdir -p $out
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; free -g | head -2) > $out/env.txt 2>&1
timeout -k 10 300 python tests/gpu_checks/precision_report.py fp32 > $out/precision_fp32.txt 2>&1; echo "precision fp32 rc=$?"; tail -8 $out/precision_fp32.txt
timeout -k 10 300 python tests/gpu_checks/precision_report.py mixed iiwa7 mixed5 > $out/precision_mixed.txt 2>&1; echo "precision mixed rc=$?"; tail -8 $out/precision_mixed.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $out/gpu_tests.txt; tail -15 $out/gpu_tests.txt
for p in fp32 mixed; do
  timeout -k 10 200 python tools/coop_sweep.py iiwa7 $p 64,1024,4096,16384,65536,262144 > $out/sweep_iiwa7_$p.txt 2>&1; echo "sweep iiwa7 $p rc=$?"; cat $out/sweep_iiwa7_$p.txt | tail -9
done
timeout -k 10 300 python tools/coop_sweep.py atlas30 fp32 64,4096,16384,65536 > $out/sweep_atlas30_fp32.txt 2>&1; echo "sweep atlas30 rc=$?"; cat $out/sweep_atlas30_fp32.txt | tail -7
timeout -k 10 200 python tools/latency.py iiwa7 fp32 > $out/latency_iiwa7.txt 2>&1; echo "latency rc=$?"; cat $out/latency_iiwa7.txt | tail -12
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; tail -c 1500 $out/bench_default.json
