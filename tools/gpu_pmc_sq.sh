#!/bin/bash
# Issue / stall breakdown of the kernels of a python script (three separate PMC passes, csv):
# usage gpu_pmc_sq.sh <tag> <script> [args]; env for the script is inherited
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VALU" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw$i -- python3 "$@" > $out/run$i.log 2>&1
  echo "pass $i rc=$?"
  find $out/raw$i -name "*counter_collection.csv" -exec cp {} $out/counters$i.csv \;
  rm -rf $out/raw$i
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob("$out/counters*.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"])); cnt[(f, k)] += 1
with open("$out/summary.txt", "w") as o:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:8]:
        n = max(c for (f, kk), c in cnt.items() if kk == k)
        o.write(f"{k} dispatches={n}\n   " + "\n   ".join(f"{c}={v/n:.5g}" for c, v in sorted(agg[k].items())) + "\n")
print(open("$out/summary.txt").read())
PY
