#!/bin/bash
set -u
out=gpurun_out/r5m; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_shapes_configs.py tests/test_gpu_resnet.py tests/test_gpu_ops.py tests/test_gpu_production_shapes.py tests/test_gpu_moco.py -m gpu -q -x --timeout 800 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for k in 0 1; do echo "== SSL4GIE_NT256_NJ2=$k"; SSL4GIE_NT256_NJ2=$k python tools/r50_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/r50_nj2_$k.log; done
BENCH_ARGS="--workload moco" bash tools/gpu_sweep.sh r5m "SSL4GIE_NT256_NJ2=0" "SSL4GIE_NT256_NJ2=1"
