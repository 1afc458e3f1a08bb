#!/bin/bash
# hd-64 attention backward at one workgroup per CU: persistent + two-batch prefetch (SSL4GIE_ATTN_PREFETCH=0 -> old kernel)
set -u
out=gpurun_out/r5z; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "attention tests rc=$rc"; tail -5 $out/tests.log
[ $rc -ne 0 ] && exit 1
C="vitb.full:256:197:12:64,det.window:64:256:12:64,n160.hd64:256:160:12:64,n224.hd64:256:224:12:64,vitb.b128:128:197:12:64,vitb.b32:32:197:12:64,vitb.b21:21:197:12:64,mae.dec:256:197:16:32"
for k in 0 1; do echo "== SSL4GIE_ATTN_PREFETCH=$k"; SSL4GIE_ATTN_PREFETCH=$k ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee $out/attn_pf$k.log; done
