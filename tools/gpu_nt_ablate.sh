#!/bin/bash
# NT 256x256 kernel: per-shape times under the epilogue ablation knobs and the pre-issue / snake options
set -u
out=gpurun_out/${1:-r03b}
mkdir -p $out
run() { echo "== $*" >> $out/nt_ablate.log; env "$@" GEMM_ITERS=30 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us %s TF/s;", $1, $2, $(NF-3), $(NF-1)} END {print ""}' >> $out/nt_ablate.log; }
run SSL4GIE_NT256_EARLY=0
run SSL4GIE_NT256_EARLY=1
run SSL4GIE_NT256_EARLY=0 SSL4GIE_NT256_NOEPI=1
run SSL4GIE_NT256_EARLY=0 SSL4GIE_NT256_NOEPI=2
run SSL4GIE_NT256_EARLY=0 SSL4GIE_NT256_NOEPI=3
run SSL4GIE_NT256_EARLY=1 SSL4GIE_NT256_NOEPI=3
run SSL4GIE_NT256_EARLY=0
run SSL4GIE_NT256_EARLY=1
cat $out/nt_ablate.log
