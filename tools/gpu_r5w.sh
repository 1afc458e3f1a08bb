#!/bin/bash
# de-phased NT launches (late workgroups spread over the XCDs): per shape (hot operands), correctness of the tile walk, whole step
set -u
out=gpurun_out/r5w; mkdir -p $out
SSL4GIE_NT_DEPHASE=50 timeout -k 10 600 python -m pytest tests/test_gpu_production_shapes.py -m gpu -q -x --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "production-shape tests with de-phasing rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit 1
for p in 0 50 100; do echo "== SSL4GIE_NT_DEPHASE=$p"; SSL4GIE_NT_DEPHASE=$p GEMM_SKIP_TN=1 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu" | tee $out/nt_$p.log; done
