#!/bin/bash
# Why do two co-resident waves per SIMD add (almost) no throughput in the generated kernels?  SQ / SQC counter passes over the
# UNSPLIT iiwa-7 forward-dynamics-gradient kernel with one wave per SIMD (K = 65536: 1024 single-wave blocks) and with two
# (K = 131072: 2048 blocks).  Counters in their own runs (--pmc with --kernel-trace only).  usage: tools/pmc_pairs.sh <outdir>
set -o pipefail
export TMPDIR=/tmp
out=$1; mkdir -p $out
rocprofv3 -L > $out/counters_list.txt 2>&1 || true
grep -o -E "\b(SQC?_[A-Z_0-9]+|GRBM_[A-Z_0-9]+)\b" $out/counters_list.txt | sort -u > $out/counters_avail.txt
for K in 65536 131072; do
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
              "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
              "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
              "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_TC_INST_REQ" \
              "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
              "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace -d $out/K${K}_p$i --output-format csv -- python3 tools/run_alg.py iiwa7 4 $K 1 1 5 > $out/K${K}_p$i.log 2>&1 || echo "K=$K pass $i failed ($ctrs)" >> $out/summary.txt
    f=$(find $out/K${K}_p$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" $K >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gradient" in k:
        print("K=%s" % sys.argv[2], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
    fi
    rm -rf $out/K${K}_p$i
  done
done
cat $out/summary.txt
