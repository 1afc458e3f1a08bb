#!/bin/bash
# Round 4: epilogue rework of gemm_nt256 (bias row by LDS-DMA, software-pipelined staging): correctness with the
# release library, per-shape times, in-kernel stamps (debug library).  usage: gpurun -- bash tools/gpu_nt_epi.sh <tag>
set -u
out=gpurun_out/${1:-r04c}
mkdir -p $out
log=$out/nt_epi.log
timeout -k 10 600 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_ops.py tests/test_gpu_conv.py -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $log
tail -3 $out/tests.log >> $log
run() { echo "== $*" >> $log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT|^TN" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $log; }
run A=1
run A=2
for w in 4 8; do
  echo "== stamps NOEPI=$w" >> $log
  SSL4GIE_NT256_NOEPI=$w timeout -k 10 200 python tools/nt_stamps.py 2>/dev/null >> $log
done
cat $log
