#!/bin/bash
set -u
out=gpurun_out/${1:-r03d}
mkdir -p $out
run() { echo "== $*" >> $out/nt_cus.log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $out/nt_cus.log; }
for cu in 240 120 60; do for ne in 0 1; do run SSL4GIE_COMPUTE_CUS=$cu SSL4GIE_NT256_NOEPI=$ne; done; done
run SSL4GIE_NT256_SC1=1
run SSL4GIE_NT256_SC1=1 SSL4GIE_NT256_SKEW_US=8
run SSL4GIE_NT256_SC1=1 SSL4GIE_NT256_SKEW_US=12
cat $out/nt_cus.log
