#!/bin/bash
# Round-3 measurements on the GPU box: GPU tests, smoke, bench lines, rocprofv3 kernel trace + HBM traffic counters, sweeps, latencies.
# Stops at the first failing step (no GPU work after a fault or a timeout).  Everything it executes is prebuilt by
# __graft_entry__.build() (libraries, harnesses, tools/ubench binaries) -- checked before the first GPU step.
set -o pipefail
export TMPDIR=/tmp
export GRID_REQUIRE_REGRESSION_LIBS=${GRID_REQUIRE_REGRESSION_LIBS:-1}
out=gpurun_out/${1:-r03f}; mkdir -p $out
stage=${2:-all}      # tests | bench | prof | sweeps | all  (one gpurun call holds at most 20 minutes: run the stages in separate calls)
want() { [ "$stage" = all ] || [ "$stage" = "$1" ]; }
for f in tools/ubench/phase_stamps tools/ubench/pair_stamps; do [ -x $f ] || { echo "missing $f: run __graft_entry__.build() first"; exit 1; }; done
step() { name=$1; shift; "$@" || { echo "$name FAILED rc=$?"; exit 1; }; echo "$name ok"; }
want tests && { step pytest  bash -c "timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/gpu_tests.txt 2>&1"; tail -4 $out/gpu_tests.txt; }
want tests && { step smoke   bash -c "timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.txt 2>&1"; tail -3 $out/smoke.txt; }
want bench && { step bench   bash -c "timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err"; }
want bench && { step bench_all bash -c "timeout -k 10 300 python bench.py --all-kernels > $out/bench_iiwa7_16384.json 2>> $out/bench.err"; }
want bench && { step bench_mixed bash -c "timeout -k 10 300 python bench.py --precision mixed --no-secondary > $out/bench_iiwa7_16384_mixed.json 2>> $out/bench.err"; }
want bench && { step bench_atlas bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 16384 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_16384.json 2>> $out/bench.err"; }
want bench && { step bench_atlas65 bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 65536 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_65536.json 2>> $out/bench.err"; }
want prof && { step kt      bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/kt.log 2>&1"; }
want prof && { step fetch   bash -c "timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1"; }
want prof && { step write   bash -c "timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1"; }
want prof && { step fetch65 bash -c "timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch65 --output-format csv -- python3 bench.py --robot atlas30 --batch 65536 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/fetch65.log 2>&1"; }
want prof && { step write65 bash -c "timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write65 --output-format csv -- python3 bench.py --robot atlas30 --batch 65536 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $out/write65.log 2>&1"; }
want sweeps && { step latency_iiwa  bash -c "timeout -k 10 200 python tools/latency.py iiwa7 fp32 > $out/latency_iiwa7.txt 2>&1"; }
want sweeps && { step latency_all_iiwa  bash -c "timeout -k 10 200 python tools/latency_all.py iiwa7 fp32 > $out/latency_all_iiwa7_fp32.txt 2>&1"; }
want sweeps && { step latency_all_atlas bash -c "timeout -k 10 200 python tools/latency_all.py atlas30 fp32 > $out/latency_all_atlas30_fp32.txt 2>&1"; }
want sweeps && { step latency_all_atlas_mixed bash -c "timeout -k 10 200 python tools/latency_all.py atlas30 mixed > $out/latency_all_atlas30_mixed.txt 2>&1"; }
want sweeps && { step latency_atlas bash -c "timeout -k 10 300 python tools/latency.py atlas30 fp32 > $out/latency_atlas30.txt 2>&1"; }
want sweeps && { step sweep_atlas bash -c "timeout -k 10 300 python tools/coop_sweep.py atlas30 fp32 64,4096,8192,16384,32768,65536,131072 > $out/sweep_atlas30_fp32.txt 2>&1"; tail -9 $out/sweep_atlas30_fp32.txt; }
want sweeps && { step sweep_iiwa  bash -c "timeout -k 10 200 python tools/coop_sweep.py iiwa7 fp32 64,1024,4096,16384,65536,262144,1048576 > $out/sweep_iiwa7_fp32.txt 2>&1"; tail -9 $out/sweep_iiwa7_fp32.txt; }
want sweeps && { step precision_fp32  bash -c "timeout -k 10 400 python tests/gpu_checks/precision_report.py fp32 > $out/precision_report_fp32.txt 2>&1"; }
want sweeps && { step precision_mixed bash -c "timeout -k 10 400 python tests/gpu_checks/precision_report.py mixed > $out/precision_report_mixed.txt 2>&1"; }
want bench && { step bench_atlas_mixed bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 16384 --precision mixed --no-secondary --steps 50 --warmup 5 > $out/bench_atlas30_16384_mixed.json 2>> $out/bench.err"; }
want sweeps && { step ksweep  bash -c "timeout -k 10 300 python tools/ksweep_all.py atlas30 fp32 64,1024,4096,16384,65536 > $out/ksweep_atlas30_fp32.txt 2>&1"; }
want sweeps && { step stamps  bash -c "timeout -k 10 120 ./tools/ubench/phase_stamps > $out/phase_stamps.txt 2>&1"; }
want sweeps && { step pairs   bash -c "timeout -k 10 200 ./tools/ubench/pair_stamps > $out/pair_stamps.txt 2>&1"; }
find $out -name "*.csv" | head -12
# the headline kernel alone (the default bench line also runs it on two streams at once, which stretches those dispatches)
want prof2 && { step kt_headline bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt_headline --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $out/kt_headline.log 2>&1"; }
exit 0
