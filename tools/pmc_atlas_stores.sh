#!/bin/bash
# Where do the 12 us go that a tile of the Atlas-30 tile-cooperative kernel loses on a full chip (K = 64: 55.5 us, K = 16384: 67.6 us)?
# SQ passes over the same kernel at 1, 64 and 256 tiles.  Counters in their own runs (--pmc with --kernel-trace only).
# usage: tools/pmc_atlas_stores.sh <outdir>
set -o pipefail
export TMPDIR=/tmp
out=$1; mkdir -p $out
for K in ${GRID_PMC_KS:-64 4096 16384 65536}; do
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
              "SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VALU" \
              "TA_BUSY_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace -d $out/K${K}_p$i --output-format csv -- python3 tools/run_alg.py atlas30 4 $K 1 0 5 2 > $out/K${K}_p$i.log 2>&1 || echo "K=$K pass $i failed ($ctrs)" >> $out/summary.txt
    f=$(find $out/K${K}_p$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" $K >> $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "gradient" in r["Kernel_Name"] and r["Counter_Name"] == rows[-1]["Counter_Name"]]
print("K=%s kernel duration under the counters [us]: min %.1f avg %.1f" % (sys.argv[2], min(dur) / 1e3, sum(dur) / len(dur) / 1e3))
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
exit 0
