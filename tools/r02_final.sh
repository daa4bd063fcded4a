#!/bin/bash
# Round-2 final measurements on the GPU box: GPU tests, smoke, bench lines, rocprofv3 kernel trace + HBM traffic counters, sweeps.
# Stops at the first failing step (no GPU work after a fault or a timeout).
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-r02f}; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/gpu_tests.txt 2>&1 || { echo "pytest FAILED rc=$?"; exit 1; }; echo "pytest ok"; tail -4 $out/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { echo "smoke FAILED rc=$?"; exit 1; }; echo "smoke ok"; tail -3 $out/smoke.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err || { echo "bench FAILED rc=$?"; exit 1; }; echo "bench ok"
timeout -k 10 300 python bench.py --all-kernels > $out/bench_iiwa7_16384.json 2>> $out/bench.err || { echo "bench all-kernels FAILED rc=$?"; exit 1; }; echo "bench all-kernels ok"
timeout -k 10 300 python bench.py --precision mixed --no-secondary > $out/bench_iiwa7_16384_mixed.json 2>> $out/bench.err || { echo "bench mixed FAILED rc=$?"; exit 1; }; echo "bench mixed ok"
timeout -k 10 300 python bench.py --robot atlas30 --batch 16384 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_16384.json 2>> $out/bench.err || { echo "bench atlas 16384 FAILED rc=$?"; exit 1; }; echo "bench atlas 16384 ok"
timeout -k 10 300 python bench.py --robot atlas30 --batch 65536 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_65536.json 2>> $out/bench.err || { echo "bench atlas 65536 FAILED rc=$?"; exit 1; }; echo "bench atlas 65536 ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/kt.log 2>&1 || { echo "kt FAILED rc=$?"; exit 1; }; echo "kt ok"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1 || { echo "fetch FAILED rc=$?"; exit 1; }; echo "fetch ok"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1 || { echo "write FAILED rc=$?"; exit 1; }; echo "write ok"
timeout -k 10 300 python tools/coop_sweep.py atlas30 fp32 64,4096,8192,16384,32768,65536,131072 > $out/sweep_atlas30_fp32.txt 2>&1 || { echo "sweep atlas30 FAILED rc=$?"; exit 1; }; echo "sweep atlas30 ok"; tail -9 $out/sweep_atlas30_fp32.txt
timeout -k 10 200 python tools/coop_sweep.py iiwa7 fp32 64,1024,4096,16384,65536,262144,1048576 > $out/sweep_iiwa7_fp32.txt 2>&1 || { echo "sweep iiwa7 FAILED rc=$?"; exit 1; }; echo "sweep iiwa7 ok"; tail -9 $out/sweep_iiwa7_fp32.txt
timeout -k 10 200 python tools/latency.py iiwa7 fp32 > $out/latency_iiwa7.txt 2>&1 || { echo "latency FAILED rc=$?"; exit 1; }; echo "latency ok"
timeout -k 10 200 python tools/latency.py atlas30 fp32 > $out/latency_atlas30.txt 2>&1 || { echo "latency atlas FAILED rc=$?"; exit 1; }; echo "latency atlas ok"
timeout -k 10 400 python tests/gpu_checks/precision_report.py fp32 > $out/precision_report_fp32.txt 2>&1 || { echo "precision fp32 FAILED rc=$?"; exit 1; }; echo "precision fp32 ok"
timeout -k 10 400 python tests/gpu_checks/precision_report.py mixed > $out/precision_report_mixed.txt 2>&1 || { echo "precision mixed FAILED rc=$?"; exit 1; }; echo "precision mixed ok"
timeout -k 10 300 python bench.py --robot atlas30 --batch 16384 --precision mixed --no-secondary --steps 50 --warmup 5 > $out/bench_atlas30_16384_mixed.json 2>> $out/bench.err || { echo "bench atlas mixed FAILED rc=$?"; exit 1; }; echo "bench atlas mixed ok"
timeout -k 10 300 python tools/ksweep_all.py atlas30 fp32 64,1024,4096,16384,65536 > $out/ksweep_atlas30_fp32.txt 2>&1 || { echo "ksweep FAILED rc=$?"; exit 1; }; echo "ksweep ok"
timeout -k 10 300 python tools/coop_sweep.py atlas30 mixed 64,4096,16384,65536 > $out/sweep_atlas30_mixed.txt 2>&1 || { echo "sweep atlas mixed FAILED rc=$?"; exit 1; }; echo "sweep atlas mixed ok"
timeout -k 10 120 ./tools/ubench/phase_stamps > $out/phase_stamps.txt 2>&1 || { echo "stamps FAILED rc=$?"; exit 1; }; echo "stamps ok"
find $out -name "*.csv" | head -12
