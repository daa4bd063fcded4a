#!/bin/bash
# Round-2 GPU session 2: full GPU test suite, smoke, Atlas sweep with the prefetched Minv product, then build-variant experiments
# (register-capped fine splits: do two waves per SIMD pay now that the kernels are spill-free?).  Experiments rebuild libraries ON
# THE BOX (its copy only), so they come last.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r02b; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > $out/gpu_tests.txt 2>&1; echo "pytest rc=$?" | tee -a $out/gpu_tests.txt; tail -12 $out/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.txt
timeout -k 10 300 python tools/coop_sweep.py atlas30 fp32 64,4096,16384,32768,65536 > $out/sweep_atlas30_fp32.txt 2>&1; echo "sweep atlas30 rc=$?"; tail -8 $out/sweep_atlas30_fp32.txt
timeout -k 10 400 python tools/exp_variants.py iiwa7 '{"experimental": {"split_cap": [2, 7]}}' 4096,16384,32768 > $out/exp_cap27.txt 2>&1; echo "exp cap27 rc=$?"; tail -4 $out/exp_cap27.txt
timeout -k 10 400 python tools/exp_variants.py iiwa7 '{"experimental": {"split_cap": [2, 3, 4, 7]}}' 4096,16384,32768,65536 > $out/exp_cap2347.txt 2>&1; echo "exp cap2347 rc=$?"; tail -5 $out/exp_cap2347.txt
timeout -k 10 400 python tools/exp_variants.py iiwa7 '{"experimental": {"split_cap": [2, 3, 4, 7]}}' 4096,16384,32768,65536 3 > $out/exp_cap2347_did.txt 2>&1; echo "exp cap2347 dID rc=$?"; tail -5 $out/exp_cap2347_did.txt
