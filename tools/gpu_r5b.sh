#!/bin/bash
# round 5: TN K-loop ablations (compile-time variants, tools/build_variant.sh)
set -u
out=gpurun_out/r5b; mkdir -p $out
for abl in 0 1 2 4 8 6 10 12 14 15; do
  echo "== ABL=$abl"
  SSL4GIE_DEBUG_LIB=xabl$abl TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py > $out/abl$abl.log 2>&1; echo "rc=$?" >> $out/abl$abl.log; grep "TN-pair" $out/abl$abl.log | sed 's/(GEMM + 2 slab reductions)//'
done
echo "== release"; TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep TN-pair | sed 's/(GEMM + 2 slab reductions)//'
