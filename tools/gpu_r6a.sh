#!/bin/bash
# round 6: table form of the GELU pair epilogue — parity, per-shape A/B, whole-step A/B
set -u
out=gpurun_out/r6a; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_gelu_table.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.log
for t in 1 0; do echo "== SSL4GIE_GELU_TABLE=$t"; SSL4GIE_GELU_TABLE=$t GEMM_SKIP_TN=1 GEMM_CASES=enc.fc1,dec.fc1,enc.qkv,dec.qkv timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee $out/nt_tab$t.log; done
bash tools/gpu_sweep.sh r6a "SSL4GIE_GELU_TABLE=1" "SSL4GIE_GELU_TABLE=0"
