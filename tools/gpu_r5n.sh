#!/bin/bash
set -u
out=gpurun_out/r5n; mkdir -p $out
SSL4GIE_ATTN_BWD1=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "attention tests (single-pass backward) rc=$rc"; tail -12 $out/tests.log
for k in 0 1; do echo "== SSL4GIE_ATTN_BWD1=$k"; SSL4GIE_ATTN_BWD1=$k ATTN_CASES="mae.dec:256:197:16:32,vitb.full:256:197:12:64,det.window:64:256:12:64,n224.hd32:256:224:16:32,n160.hd64:256:160:12:64" python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee $out/attn_$k.log; done
