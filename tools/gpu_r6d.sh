#!/bin/bash
# round 6: 2 x 2 bilinear kernels — parity, per-map A/B, depth step A/B
set -u
out=gpurun_out/r6d; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_dpt.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -5 | tee $out/tests.log
for v in 1 0; do echo "== SSL4GIE_BILINEAR22=$v"; SSL4GIE_BILINEAR22=$v timeout -k 10 200 python tools/experiments/bilinear_bench.py 2>&1 | grep -v amdgpu | tee -a $out/bilinear_ab.log; done
BENCH_ARGS="--workload depth" bash tools/gpu_sweep.sh r6d "SSL4GIE_BILINEAR22=1" "SSL4GIE_BILINEAR22=0"
