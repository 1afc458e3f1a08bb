#!/bin/bash
# s_setprio policy of the GELU epilogues (debug library): usage: gpurun -- bash tools/gpu_nt_prio.sh <tag>
set -u
out=gpurun_out/${1:-r04j}; mkdir -p $out; log=$out/nt_prio.log
export SSL4GIE_DEBUG_LIB=1 GEMM_SKIP_TN=1
run() { echo "== $*" >> $log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $log; }
for rep in 1 2; do for p in 0 1 2 3; do run SSL4GIE_NT256_PRIO=$p; done; done
for p in 0 1 2 3; do for w in 4 8; do
  echo "== stamps PRIO=$p NOEPI=$w" >> $log
  SSL4GIE_NT256_PRIO=$p SSL4GIE_NT256_NOEPI=$w timeout -k 10 200 python tools/nt_stamps.py 2>/dev/null | grep fc1 >> $log
done; done
cat $log
