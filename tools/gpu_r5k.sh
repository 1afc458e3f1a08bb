#!/bin/bash
set -u
out=gpurun_out/r5k; mkdir -p $out
SSL4GIE_WGRAD_STREAMS=2 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_mae.py -m gpu -q -x --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "tests (2 streams) rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_sweep.sh r5k "X=0" "SSL4GIE_WGRAD_STREAMS=2" "SSL4GIE_WGRAD_STREAMS=2 SSL4GIE_TN_FILL=50" "SSL4GIE_WGRAD_STREAMS=2 SSL4GIE_TN_FILL=35" "SSL4GIE_TN_FILL=50" "SSL4GIE_WGRAD_STREAMS=2 SSL4GIE_TN_FILL=100"
