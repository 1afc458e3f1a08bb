#!/bin/bash
# PMC passes (separate runs, csv): usage gpu_pmc.sh <tag> <python script> [args]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw$i -- python3 "$@" > $out/run$i.log 2>&1
  echo "pass $i rc=$?" >> $out/summary.txt
  find $out/raw$i -name "*counter_collection.csv" -exec cp {} $out/counters$i.csv \;
  rm -rf $out/raw$i
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in sorted(glob.glob("$out/counters*.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[(f, k)] += 1
names = sorted(agg, key=lambda k: -agg[k].get("FETCH_SIZE", 0))
with open("$out/pmc_summary.txt", "w") as o:
    for k in names[:25]:
        n = max(c for (f, kk), c in cnt.items() if kk == k)
        o.write(f"{k} dispatches={n} " + " ".join(f"{c}={v/n:.4g}" for c, v in sorted(agg[k].items())) + "\n")
print(open("$out/pmc_summary.txt").read())
PY
