#!/bin/bash
set -u
out=gpurun_out/${1:-r03c}
mkdir -p $out
run() { echo "== $*" >> $out/nt_skew.log; env "$@" GEMM_ITERS=30 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $out/nt_skew.log; }
for sk in 0 4 8 12 16 24; do run SSL4GIE_NT256_SKEW_US=$sk; done
cat $out/nt_skew.log
