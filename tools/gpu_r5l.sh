#!/bin/bash
set -u
out=gpurun_out/r5l; mkdir -p $out
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_mae.py tests/test_gpu_reference_loop.py tests/test_gpu_optim.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_sweep.sh r5l "SSL4GIE_LN_SIDE=0" "SSL4GIE_LN_SIDE=1"
