#!/bin/bash
# End-of-round measurements on the GPU box: GPU tests, smoke, benches, rocprofv3 kernel trace + HBM traffic counters.
# usage: bash tools/final_round_profile.sh <outdir under gpurun_out>
set -o pipefail
export TMPDIR=/tmp
out=$1; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo "rc=$?" >> $out/gpu_tests.txt; tail -3 $out/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "rc=$?" >> $out/smoke.txt; tail -2 $out/smoke.txt
timeout -k 10 300 python bench.py --all-kernels > $out/bench_iiwa7_16384.json 2> $out/bench_iiwa7.err; echo "bench iiwa rc=$?"
timeout -k 10 400 python bench.py --robot atlas30 --batch 65536 --all-kernels --steps 50 --warmup 5 > $out/bench_atlas30_65536.json 2> $out/bench_atlas.err; echo "bench atlas rc=$?"
timeout -k 10 400 python bench.py --robot atlas30 --batch 32768 --all-kernels --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_atlas30_32768.json 2>> $out/bench_atlas.err; echo "bench atlas 32k rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1; echo "write rc=$?"
timeout -k 10 300 python tools/ksweep.py iiwa7 4096 16384 65536 262144 1048576 > $out/ksweep_iiwa7.txt 2>&1; echo "ksweep rc=$?"
find $out -name "*.csv" | head -20
