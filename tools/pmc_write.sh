#!/bin/bash
# HBM write path of one forward-dynamics-gradient variant: WRITE_SIZE / FETCH_SIZE and the TCC->EA write request mix, one counter
# pass each (rocprofv3 --pmc with --kernel-trace only).  usage: tools/pmc_write.sh <robot or variant> <K> <coop mode> <outdir>
set -o pipefail
export TMPDIR=/tmp
robot=$1; K=$2; coop=$3; out=$4; mkdir -p $out
i=0
for ctrs in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace -d $out/p$i --output-format csv -- python3 tools/run_alg.py $robot 4 $K 1 0 4 $coop > $out/p$i.log 2>&1 || { echo "pass $i ($ctrs) failed" >> $out/summary.txt; continue; }
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$robot K=$K" >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gradient" in k:
        print(sys.argv[2], k, {c: sum(v) / len(v) for c, v in d.items()})
PY
  fi
  rm -rf $out/p$i
done
cat $out/summary.txt
