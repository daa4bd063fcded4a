#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02e; mkdir -p $out
timeout -k 10 120 ./tools/ubench/phase_stamps > $out/phase_stamps.txt 2>&1; echo "stamps rc=$?"; cat $out/phase_stamps.txt
timeout -k 10 300 python tools/coop_sweep.py atlas30 fp32 64,4096,8192,16384,32768,65536 > $out/sweep_atlas30_fp32.txt 2>&1; echo "sweep atlas30 rc=$?"; tail -8 $out/sweep_atlas30_fp32.txt
timeout -k 10 200 python tools/coop_sweep.py iiwa7 fp32 64,4096,16384,65536 > $out/sweep_iiwa7_fp32.txt 2>&1; echo "sweep iiwa7 rc=$?"; tail -6 $out/sweep_iiwa7_fp32.txt
timeout -k 10 600 python -m pytest tests/test_coop.py tests/test_gpu_parity.py -m gpu -x -q > $out/gpu_tests_coop.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/gpu_tests_coop.txt
