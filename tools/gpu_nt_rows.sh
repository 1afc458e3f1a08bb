#!/bin/bash
# Feasibility of a row-interleaved schedule (debug library): K-loop time with the second wave row idle / busy
# with GELU-sized VALU work.  usage: gpurun -- bash tools/gpu_nt_rows.sh <tag>
set -u
out=gpurun_out/${1:-r04k}; mkdir -p $out; log=$out/nt_rows.log
export SSL4GIE_DEBUG_LIB=1 GEMM_SKIP_TN=1
run() { echo "== $*" >> $log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $log; }
for rep in 1 2; do for p in 1 9 10; do run SSL4GIE_NT256_NOEPI=$p; done; done
cat $log
