#!/bin/bash
set -u
out=gpurun_out/r5j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_production_shapes.py -m gpu -q -x -k "tn or weight or wgrad or pair or group" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for k in 4 1 4 1; do
  echo "== SSL4GIE_TN256K=$k"
  SSL4GIE_TN256K=$k TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep TN-pair | sed 's/(GEMM + 2 slab reductions)//' | tee -a $out/pair_k$k.log
done
bash tools/gpu_sweep.sh r5j "SSL4GIE_TN256K=4" "SSL4GIE_TN256K=1"
