#!/bin/bash
# effective clock (GRBM_GUI_ACTIVE / 8 / duration) of the GEMM kernels in the MAE step at two grid sizings
set -u
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $out
cd $GRAFT_REPO_ROOT
for cus in 240 208; do
  export SSL4GIE_COMPUTE_CUS=$cus
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/raw$cus -- python3 bench.py --steps 3 --warmup 2 --prof-steps 0 --no-cpu-baseline > $out/run$cus.log 2>&1
  echo "cus $cus rc=$?"
  find $out/raw$cus -name "*counter_collection.csv" -exec cp {} $out/counters$cus.csv \;
  find $out/raw$cus -name "*kernel_trace.csv" -exec cp {} $out/trace$cus.csv \;
  rm -rf $out/raw$cus
  python3 - <<PY
import csv, collections
dur = {}
for r in csv.DictReader(open("$out/trace$cus.csv")):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"][:48], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open("$out/counters$cus.csv")):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k, d = dur.get(r["Dispatch_Id"], (None, 0))
    if k is None: continue
    a = agg[k]; a[0] += float(r["Counter_Value"]); a[1] += d; a[2] += 1
print("COMPUTE_CUS=$cus")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {k:48s} n={a[2]:4d} total {a[1]/1e6:8.2f} ms  clock {a[0]/8/a[1]:.3f} GHz")
PY
done
