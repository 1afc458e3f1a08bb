#!/bin/bash
# 2-phase K-tile schedule of gemm_nt256 (debug library: SSL4GIE_NT256_PH2=0/1): exact tests, per-shape times, K-loop only
set -u
out=gpurun_out/${1:-r04r}; mkdir -p $out; log=$out/nt_ph2.log
export SSL4GIE_DEBUG_LIB=1
SSL4GIE_NT256_PH2=1 timeout -k 10 600 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider > $out/tests_ph2.log 2>&1; echo "tests ph2 rc=$?" | tee -a $log
tail -2 $out/tests_ph2.log >> $log
export GEMM_SKIP_TN=1
run() { echo "== $*" >> $log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $log; }
for rep in 1 2; do for p in 0 1; do run SSL4GIE_NT256_PH2=$p; done; done
for p in 0 1; do run SSL4GIE_NT256_PH2=$p SSL4GIE_NT256_NOEPI=1; done
cat $log
