#!/bin/bash
# round 5, first measurement: baseline bench line of this box, TN pair launches hot vs cold, TN K-loop ablations
set -u
out=gpurun_out/r5a; mkdir -p $out
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $out/box.txt
echo "== bench" ; timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $out/bench.log 2>&1; echo "rc=$?" >> $out/bench.log; tail -2 $out/bench.log
echo "== tn pair (release)"; timeout -k 10 300 python tools/tn_pair_bench.py > $out/tn_pair.log 2>&1; echo "rc=$?" >> $out/tn_pair.log; cat $out/tn_pair.log
for abl in 0 1 2 4 8 6 14 15; do
  echo "== tn pair cold, debug lib, ABL=$abl"
  SSL4GIE_DEBUG_LIB=1 SSL4GIE_TN256_ABL=$abl TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py > $out/tn_pair_abl$abl.log 2>&1; echo "rc=$?" >> $out/tn_pair_abl$abl.log; grep "TN-pair" $out/tn_pair_abl$abl.log
done
echo "== gemm bench"; timeout -k 10 300 python tools/gemm_bench.py > $out/gemm_bench.log 2>&1; echo "rc=$?" >> $out/gemm_bench.log; cat $out/gemm_bench.log
