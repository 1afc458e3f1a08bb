#!/bin/bash
set -u
out=gpurun_out/r5za; mkdir -p $out
SSL4GIE_ATTN_PREFETCH32=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "attention tests (hd32 prefetch) rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit 1
C="mae.dec:256:197:16:32,n224.hd32:256:224:16:32,n160.hd32:256:160:16:32,n256.hd32:64:256:16:32"
for k in 0 1; do echo "== SSL4GIE_ATTN_PREFETCH32=$k"; SSL4GIE_ATTN_PREFETCH32=$k ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee $out/attn_pf32_$k.log; done
