#!/bin/bash
# LDS bank-conflict counters for a python script: usage gpu_pmc_lds.sh <tag> <script> [args]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmclds_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/raw -- python3 "$@" > $out/run.log 2>&1
echo "rc=$?" >> $out/run.log
find $out/raw -name "*counter_collection.csv" -exec cp {} $out/counters.csv \;
rm -rf $out/raw
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen=set()
for r in csv.DictReader(open("$out/counters.csv")):
    k = r["Kernel_Name"][:80]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] += 1
with open("$out/summary.txt", "w") as o:
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_LDS", 0))[:14]:
        a = agg[k]; n = cnt[k]
        act = a.get("SQ_LDS_IDX_ACTIVE", 0)
        o.write(f"{k} dispatches={n} " + " ".join(f"{c}={v/n:.4g}" for c, v in sorted(a.items())) +
                (f" conflict/active={a.get('SQ_LDS_BANK_CONFLICT',0)/act:.3f}" if act else "") + "\n")
print(open("$out/summary.txt").read())
PY
