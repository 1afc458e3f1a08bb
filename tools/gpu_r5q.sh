#!/bin/bash
# attention backward, persistent over heads with the next head's operands prefetched into registers
set -u
out=gpurun_out/r5q; mkdir -p $out
C="mae.enc:256:50:12:64,mae.dec:256:197:16:32,vitb.full:256:197:12:64,det.window:64:256:12:64,n224.hd32:256:224:16:32,n160.hd64:256:160:12:64,n100.hd64:256:100:12:64,n128.hd32:256:128:16:32"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "attention tests rc=$rc"; tail -5 $out/tests.log
[ $rc -ne 0 ] && exit 1
echo "== base (HEAD)"; SSL4GIE_DEBUG_LIB=xbase ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee $out/attn_base.log
for k in 0 1; do echo "== SSL4GIE_ATTN_PERSIST=$k"; SSL4GIE_ATTN_PERSIST=$k ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee $out/attn_p$k.log; done
