#!/bin/bash
set -u
out=gpurun_out/${1:-r03f}
mkdir -p $out
timeout -k 10 300 python tools/ntw_check.py > $out/check.log 2>&1
python - <<'PY' $out/check.log
import sys, json, re
for ln in open(sys.argv[1]):
    m = re.match(r"NT256W=(\d) (\{.*\})", ln)
    if not m: print(ln[:300]); continue
    d = json.loads(m.group(2))
    print("W=" + m.group(1), {k: {a: (round(b, 5) if not isinstance(b, list) else [round(x, 5) for x in b]) if not isinstance(b, bool) else b for a, b in v.items()} for k, v in d.items() if k in ("12800x768x768", "50432x512x512", "4096x2048x512")})
PY
run() { echo "== $*" >> $out/ntw.log; env "$@" GEMM_ITERS=30 timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -E "^NT|rror" | awk '{printf "%s %s %s us %s TF/s;", $1, $2, $(NF-3), $(NF-1)} END {print ""}' >> $out/ntw.log; }
run SSL4GIE_NT256W=0
run SSL4GIE_NT256W=1
run SSL4GIE_NT256W=1 SSL4GIE_NT256_NOEPI=1
run SSL4GIE_NT256W=0
run SSL4GIE_NT256W=1
cat $out/ntw.log
