#!/bin/bash
# Round-2 GPU session 4: GPU test suite, sweeps with the hoisting cooperative cores (small robots), SQ counters of the Atlas-30 variants.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02d; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $out/gpu_tests.txt; tail -6 $out/gpu_tests.txt
timeout -k 10 200 python tools/coop_sweep.py iiwa7 fp32 64,1024,4096,16384,32768,65536,262144 > $out/sweep_iiwa7_fp32.txt 2>&1; echo "sweep iiwa7 rc=$?"; tail -9 $out/sweep_iiwa7_fp32.txt
timeout -k 10 200 python tools/coop_sweep.py iiwa7 mixed 64,4096,16384,65536 > $out/sweep_iiwa7_mixed.txt 2>&1; echo "sweep iiwa7 mixed rc=$?"; tail -6 $out/sweep_iiwa7_mixed.txt
timeout -k 10 200 python tools/latency.py iiwa7 fp32 > $out/latency_iiwa7.txt 2>&1; echo "latency rc=$?"; tail -12 $out/latency_iiwa7.txt
timeout -k 10 500 bash tools/pmc_sq.sh atlas30 16384 4 1 $out/sq_atlas_split4 > $out/sq_atlas_split4.log 2>&1; echo "sq split4 rc=$?"; tail -6 $out/sq_atlas_split4/summary.txt
timeout -k 10 500 bash tools/pmc_sq.sh atlas30 16384 0 2 $out/sq_atlas_coop > $out/sq_atlas_coop.log 2>&1; echo "sq coop rc=$?"; tail -6 $out/sq_atlas_coop/summary.txt
timeout -k 10 500 bash tools/pmc_sq.sh iiwa7 16384 4 1 $out/sq_iiwa_split4 > $out/sq_iiwa_split4.log 2>&1; echo "sq iiwa rc=$?"; tail -6 $out/sq_iiwa_split4/summary.txt
