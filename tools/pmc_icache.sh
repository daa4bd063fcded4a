#!/bin/bash
# Instruction-fetch / wait counters of the Atlas-30 dID kernel at 512 and 1024 waves.  usage: tools/pmc_icache.sh <outdir>
set -o pipefail
export TMPDIR=/tmp
out=$1; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TA_[A-Z_0-9]+)\b" | sort -u > $out/counters_avail.txt
grep -E "IFETCH|ICACHE|INST_REQ|SQC_TC|WAIT|DCACHE" $out/counters_avail.txt > $out/counters_interesting.txt
for K in 32768 65536; do
  i=0
  for ctrs in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQ_WAIT_ANY" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $ctrs --kernel-trace -d $out/K${K}_p$i --output-format csv -- python3 tools/run_alg.py atlas30 3 $K 1 1 3 > $out/K${K}_p$i.log 2>&1 || echo "K=$K pass $i failed ($ctrs)" >> $out/summary.txt
    f=$(find $out/K${K}_p$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" $K >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gradient" in k:
        print("K=%s" % sys.argv[2], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
    fi
  done
done
cat $out/summary.txt
