#!/bin/bash
# GELU-pair epilogue: lane exchange (release default) vs LDS transposition (debug library, SSL4GIE_NT256_PRIO=4)
set -u
out=gpurun_out/${1:-r04m}; mkdir -p $out; log=$out/nt_swap.log
timeout -k 10 600 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $log
tail -2 $out/tests.log >> $log
export SSL4GIE_DEBUG_LIB=1 GEMM_SKIP_TN=1
run() { echo "== $*" >> $log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $log; }
for rep in 1 2; do for p in 0 4; do run SSL4GIE_NT256_PRIO=$p; done; done
for p in 0 4; do for w in 4 8; do
  echo "== stamps PRIO=$p NOEPI=$w" >> $log
  SSL4GIE_NT256_PRIO=$p SSL4GIE_NT256_NOEPI=$w timeout -k 10 200 python tools/nt_stamps.py 2>/dev/null | grep fc1 >> $log
done; done
cat $log
