#!/bin/bash
# 256x256 GEMM kernels bring-up: correctness (default heuristic and forced), then per-shape timings
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm_bf16" --timeout 200 -p no:cacheprovider > gpurun_out/g256_tests.log 2>&1
rc=$?; echo "tests(default) rc=$rc"; tail -5 gpurun_out/g256_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo HANG; exit 1; fi
SSL4GIE_NT256=1 SSL4GIE_TN256=1 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm_bf16 or block_stack" --timeout 200 -p no:cacheprovider > gpurun_out/g256_tests_forced.log 2>&1
rc=$?; echo "tests(forced) rc=$rc"; tail -5 gpurun_out/g256_tests_forced.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo HANG; exit 1; fi
SSL4GIE_NT256=0 SSL4GIE_TN256=0 timeout -k 10 200 python tools/gemm_bench.py > gpurun_out/gemm_bench_128.log 2>&1; echo "bench128 rc=$?"
timeout -k 10 200 python tools/gemm_bench.py > gpurun_out/gemm_bench_256.log 2>&1; echo "bench256 rc=$?"
paste <(grep "^NT\|^TN" gpurun_out/gemm_bench_128.log | cut -c1-75) <(grep "^NT\|^TN" gpurun_out/gemm_bench_256.log | cut -c45-75)
