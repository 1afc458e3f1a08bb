#!/bin/bash
set -u
out=gpurun_out/r5e; mkdir -p $out
for abl in 10 26 16; do
  echo "== k-split ABL=$abl"
  SSL4GIE_DEBUG_LIB=xabl$abl TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep "TN-pair" | sed 's/(GEMM + 2 slab reductions)//' | tee $out/abl$abl.log
done
echo "== release"; TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep TN-pair | sed 's/(GEMM + 2 slab reductions)//' | tee $out/rel.log
