#!/bin/bash
set -u
out=gpurun_out/r5d; mkdir -p $out
echo "== release k-split"; TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep TN-pair | sed 's/(GEMM + 2 slab reductions)//' | tee $out/rel.log
for abl in 2 4 8 6 10 12 14; do
  echo "== k-split ABL=$abl"
  SSL4GIE_DEBUG_LIB=xabl$abl TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep "TN-pair" | sed 's/(GEMM + 2 slab reductions)//' | tee $out/abl$abl.log
done
