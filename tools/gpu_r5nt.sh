#!/bin/bash
# non-temporal epilogue stores (xnt1) / stores + residual and aux loads (xnt3) in the NT kernel: per shape and whole step
set -u
out=gpurun_out/r5nt; mkdir -p $out
for lib in "" xnt1 xnt3; do echo "== lib=$lib"; SSL4GIE_DEBUG_LIB=$lib GEMM_SKIP_TN=1 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee $out/nt_${lib:-rel}.log; done
bash tools/gpu_sweep.sh r5nt "SSL4GIE_DEBUG_LIB=" "SSL4GIE_DEBUG_LIB=xnt1" "SSL4GIE_DEBUG_LIB=xnt3"
