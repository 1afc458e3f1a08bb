#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $out
cd $GRAFT_REPO_ROOT
export SSL4GIE_DEBUG_LIB=xbase
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw$i -- python3 tools/attn_one.py > $out/run$i.log 2>&1
  echo "pass $i rc=$?"
  find $out/raw$i -name "*counter_collection.csv" -exec cp {} $out/counters$i.csv \;
  find $out/raw$i -name "*kernel_trace.csv" -exec cp {} $out/trace$i.csv \;
  rm -rf $out/raw$i
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob("$out/counters*.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"])); cnt[(f, k)] += 1
dur = collections.defaultdict(list)
for r in csv.DictReader(open("$out/trace1.csv")):
    dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in agg:
    if "attn" not in k: continue
    n = max(c for (f, kk), c in cnt.items() if kk == k)
    print(k, "dispatches", n, "avg ns", sum(dur[k]) / max(1, len(dur[k])))
    for c, v in sorted(agg[k].items()): print("   ", c, f"{v/n:.5g}")
PY
