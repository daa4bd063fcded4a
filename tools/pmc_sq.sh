#!/bin/bash
# SQ counter passes over one forward-dynamics-gradient variant.  usage: tools/pmc_sq.sh <robot> <K> <split> <coop 1|2> <outdir>
set -o pipefail
export TMPDIR=/tmp
robot=$1; K=$2; split=$3; coop=$4; out=$5; mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $ctrs --kernel-trace -d $out/p$i --output-format csv -- python3 tools/run_alg.py $robot ${ALG:-4} $K 1 $split 3 $coop > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/summary.txt
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gradient" in k:
        print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
  fi
done
cat $out/summary.txt
