#!/bin/bash
# Round-4 measurements on the GPU box, one stage per gpurun call (a call holds at most 20 minutes): GPU tests + smoke, bench lines,
# rocprofv3 kernel trace + HBM traffic counters, sweeps / latencies / precision reports.  Stops at the first failing step (no GPU work
# after a fault or a timeout).  Everything executed here is prebuilt by __graft_entry__.build().
#   usage: tools/r04_run.sh <dir under gpurun_out> tests|bench|prof|prof2|sweeps|lean
set -o pipefail
export TMPDIR=/tmp
export GRID_REQUIRE_REGRESSION_LIBS=${GRID_REQUIRE_REGRESSION_LIBS:-1}
out=gpurun_out/${1:-r04}; mkdir -p $out
stage=${2:-tests}
want() { [ "$stage" = "$1" ]; }
step() { name=$1; shift; "$@" || { echo "$name FAILED rc=$?"; exit 1; }; echo "$name ok"; }
want tests && { step pytest  bash -c "timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/gpu_tests.txt 2>&1"; tail -4 $out/gpu_tests.txt; }
want tests && { step smoke   bash -c "timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.txt 2>&1"; tail -3 $out/smoke.txt; }
want bench && { step bench   bash -c "timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench.err"; }
want bench && { step bench_all bash -c "timeout -k 10 300 python bench.py --all-kernels --no-secondary > $out/bench_iiwa7_16384.json 2>> $out/bench.err"; }
want bench && { step bench_mixed bash -c "timeout -k 10 300 python bench.py --precision mixed --no-secondary > $out/bench_iiwa7_16384_mixed.json 2>> $out/bench.err"; }
want bench && { step bench_atlas bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 16384 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_16384.json 2>> $out/bench.err"; }
want bench && { step bench_atlas65 bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 65536 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_65536.json 2>> $out/bench.err"; }
want bench && { step bench_atlas131 bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 131072 --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_atlas30_131072.json 2>> $out/bench.err"; }
want bench && { step bench_quad bash -c "timeout -k 10 300 python bench.py --robot quad12 --batch 16384 --all-kernels --steps 50 --warmup 5 > $out/bench_quad12_16384.json 2>> $out/bench.err"; }
want bench && { step bench_atlas_mixed bash -c "timeout -k 10 300 python bench.py --robot atlas30 --batch 16384 --precision mixed --no-secondary --steps 50 --warmup 5 > $out/bench_atlas30_16384_mixed.json 2>> $out/bench.err"; }
want prof && { step kt      bash -c "timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/kt.log 2>&1"; }
# HBM traffic counters: one workload per pass (tools/pmc_traffic.py keys an entry by robot, batch and kernel), FETCH_SIZE and WRITE_SIZE
# in separate passes, --kernel-trace only next to --pmc
pmc() { name=$1; ctr=$2; shift 2; step pmc_${ctr}_$name bash -c "timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace -d $out/pmc_${ctr}_$name --output-format csv -- $* > $out/pmc_${ctr}_$name.log 2>&1"; }
for ctr in FETCH_SIZE WRITE_SIZE; do
want prof && pmc iiwa7_16384 $ctr python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary
want prof && pmc atlas30_16384 $ctr python3 bench.py --robot atlas30 --batch 16384 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary
want prof && pmc atlas30_65536 $ctr python3 bench.py --robot atlas30 --batch 65536 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary
want prof && pmc atlas30_dID_16384 $ctr python3 tools/run_alg.py atlas30 3 16384 1 0 10
want prof && pmc atlas30_dID_65536 $ctr python3 tools/run_alg.py atlas30 3 65536 1 0 6
done
want prof2 && { step kt_headline bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt_headline --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $out/kt_headline.log 2>&1"; }
kt() { name=$1; shift; step kt_$name bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt_$name --output-format csv -- $* > $out/kt_$name.log 2>&1"; }
want prof2 && kt atlas30_16384 python3 bench.py --robot atlas30 --batch 16384 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary
want prof2 && kt atlas30_65536 python3 bench.py --robot atlas30 --batch 65536 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary
want prof2 && kt atlas30_dID_16384 python3 tools/run_alg.py atlas30 3 16384 1 0 300
want prof2 && kt atlas30_dID_65536 python3 tools/run_alg.py atlas30 3 65536 1 0 100
want sweeps && { step lean_id_sweep bash -c "timeout -k 10 300 python tools/lean_id_sweep.py atlas30 > $out/lean_id_sweep.txt 2>&1"; }
want sweeps && { step wave_errors bash -c "timeout -k 10 300 python tests/gpu_checks/wave_small_batch_errors.py > $out/wave_small_batch_errors.txt 2>&1"; }
want sweeps && { step precision_fp32  bash -c "timeout -k 10 400 python tests/gpu_checks/precision_report.py fp32 > $out/precision_report_fp32.txt 2>&1"; }
want sweeps && { step precision_mixed bash -c "timeout -k 10 400 python tests/gpu_checks/precision_report.py mixed > $out/precision_report_mixed.txt 2>&1"; }
want sweeps && { step latency_all_atlas bash -c "timeout -k 10 200 python tools/latency_all.py atlas30 fp32 > $out/latency_all_atlas30_fp32.txt 2>&1"; }
want sweeps && { step latency_all_quad bash -c "timeout -k 10 200 python tools/latency_all.py quad12 fp32 > $out/latency_all_quad12_fp32.txt 2>&1"; }
want sweeps && { step latency_atlas bash -c "timeout -k 10 300 python tools/latency.py atlas30 fp32 > $out/latency_atlas30.txt 2>&1"; }
want sweeps && { step sweep_atlas bash -c "timeout -k 10 300 python tools/coop_sweep.py atlas30 fp32 64,4096,8192,16384,32768,65536,131072 > $out/sweep_atlas30_fp32.txt 2>&1"; tail -9 $out/sweep_atlas30_fp32.txt; }
want sweeps && { step sweep_quad  bash -c "timeout -k 10 200 python tools/coop_sweep.py quad12 fp32 64,1024,4096,16384,65536,262144 > $out/sweep_quad12_fp32.txt 2>&1"; tail -8 $out/sweep_quad12_fp32.txt; }
want lean && { step lean_sweep bash -c "timeout -k 10 300 python tools/lean_sweep.py atlas30 64,1024,4096,16384,32768,65536,131072 > $out/lean_sweep.txt 2>&1"; cat $out/lean_sweep.txt; }
want lean && { step lean_probe bash -c "timeout -k 10 300 python tools/lean_probe.py 64,4096,16384,32768 > $out/lean_probe.txt 2>&1"; cat $out/lean_probe.txt; }
find $out -name "*.csv" | head -12
exit 0
