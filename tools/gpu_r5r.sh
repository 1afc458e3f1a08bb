#!/bin/bash
# attention backward ablations: 16 no phases (operand staging + zero stores only), 32 no operand loads, 48 both
set -u
out=gpurun_out/r5r; mkdir -p $out
C="mae.dec:256:197:16:32,vitb.full:256:197:12:64,mae.enc:256:50:12:64"
for p in 0 1; do for lib in "" xa16 xa32 xa48; do echo "== persist=$p lib=$lib"; SSL4GIE_ATTN_PERSIST=$p SSL4GIE_DEBUG_LIB=$lib ATTN_CASES=$C python tools/attn_bench.py 2>&1 | grep -v "amdgpu\|DEBUG" | tee $out/attn_p${p}_${lib:-rel}.log; done; done
