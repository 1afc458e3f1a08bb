#!/bin/bash
# lockstep test: every other NT workgroup starts 7 / 15 / 30 us late (timing only)
set -u
out=gpurun_out/r5v; mkdir -p $out
for lib in "" xst700 xst1500 xst3000; do echo "== lib=$lib"; SSL4GIE_DEBUG_LIB=$lib GEMM_SKIP_TN=1 GEMM_CASES=dec.proj,dec.fc2,dec.fc1,dec.qkv,dec.dqkv,enc.fc1 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee $out/nt_${lib:-rel}.log; done
