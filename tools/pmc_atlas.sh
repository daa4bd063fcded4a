#!/bin/bash
# PMC passes over the Atlas-30 gradient kernels (single-kernel variants).  usage: tools/pmc_atlas.sh <alg 3|4> <outdir>
set -o pipefail
export TMPDIR=/tmp
alg=$1; out=$2; mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_FLAT" "WRITE_SIZE" "FETCH_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $ctrs --kernel-trace -d $out/p$i --output-format csv -- python3 tools/run_alg.py atlas30 $alg 65536 1 1 3 > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/summary.txt
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$ctrs" >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gradient" in k:
        print(k, {c: sum(v) / len(v) for c, v in d.items()}, "dispatches", {c: len(v) for c, v in d.items()})
PY
  fi
done
cat $out/summary.txt
