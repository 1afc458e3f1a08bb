#!/bin/bash
# hd-32 attention backward with one tile per wave (14 waves per workgroup at N = 197, 64 VGPRs: 7 waves per SIMD)
set -u
out=gpurun_out/r5o; mkdir -p $out
C="mae.dec:256:197:16:32,n224.hd32:256:224:16:32,n160.hd32:256:160:16:32,n256.hd32:64:256:16:32,vitb.full:256:197:12:64"
for lib in "" xw7; do echo "== lib=$lib"; SSL4GIE_DEBUG_LIB=$lib ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee $out/attn_${lib:-rel}.log; done
SSL4GIE_DEBUG_LIB=xw7 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "attention tests (wide) rc=$rc"; tail -5 $out/tests.log
