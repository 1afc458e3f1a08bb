#!/bin/bash
# Round 4: LDS-DMA ownership of gemm_nt256 (ROLE 0 = every wave, 1 = the wr = 1 waves only) with the ablation
# modes of the debug library.  usage: gpurun -- bash tools/gpu_nt_role.sh <tag>
set -u
out=gpurun_out/${1:-r04a}
mkdir -p $out
export SSL4GIE_DEBUG_LIB=1 GEMM_SKIP_TN=1
log=$out/nt_role.log
run() { echo "== $*" >> $log; env "$@" GEMM_ITERS=20 timeout -k 10 200 python tools/gemm_bench.py 2>/dev/null | grep -E "^NT" | awk '{printf "%s %s %s us;", $1, $2, $(NF-3)} END {print ""}' >> $log; }
# correctness first: the exact-integer production shapes and the GEMM op tests on the loader-wave variant
SSL4GIE_NT256_ROLE=1 timeout -k 10 400 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_ops.py -m gpu -q -x -k "gemm or production or linear" -p no:cacheprovider > $out/tests_role1.log 2>&1; echo "tests role1 rc=$?" | tee -a $log
for role in 0 1; do
  run SSL4GIE_NT256_ROLE=$role
  run SSL4GIE_NT256_ROLE=$role SSL4GIE_NT256_NOEPI=1
  run SSL4GIE_NT256_ROLE=$role SSL4GIE_NT256_NOEPI=5
  run SSL4GIE_NT256_ROLE=$role SSL4GIE_NT256_NOEPI=6
done
run SSL4GIE_NT256_ROLE=0
run SSL4GIE_NT256_ROLE=1
run SSL4GIE_NT256_ROLE=0 SSL4GIE_NT256_NOEPI=7
run SSL4GIE_NT256_ROLE=0 SSL4GIE_NT256_NOEPI=7 SSL4GIE_COMPUTE_CUS=60
run SSL4GIE_NT256_ROLE=1 SSL4GIE_COMPUTE_CUS=60
run SSL4GIE_NT256_ROLE=0 SSL4GIE_COMPUTE_CUS=60
for role in 0 1; do for w in 4 8; do
  echo "== stamps ROLE=$role NOEPI=$w" >> $out/nt_stamps.log
  SSL4GIE_NT256_ROLE=$role SSL4GIE_NT256_NOEPI=$w timeout -k 10 200 python tools/nt_stamps.py 2>/dev/null >> $out/nt_stamps.log
done; done
cat $log $out/nt_stamps.log
tail -5 $out/tests_role1.log
