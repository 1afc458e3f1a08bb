#!/bin/bash
# compile-time ablations of the whole-head attention backward (ATTN_ABL bits: 1 no exp, 2 no score MFMAs, 4 fragment
# reads hoisted, 8 no gradient MFMAs); timing only
set -u
out=gpurun_out/r5p; mkdir -p $out
C="mae.dec/8:32:197:16:32,mae.dec:256:197:16:32,vitb.full:256:197:12:64"
for lib in "" xa1 xa2 xa3 xa4 xa8 xa15; do echo "== lib=$lib"; SSL4GIE_DEBUG_LIB=$lib ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee $out/attn_${lib:-rel}.log; done
