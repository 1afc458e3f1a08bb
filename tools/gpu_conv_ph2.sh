#!/bin/bash
# 2-phase schedule on the gathered-operand / statistics variants: conv + resnet + dpt tests (release), then MoCo / depth
# steps with SSL4GIE_NT256_PH2=0/1 in the debug library
set -u
out=gpurun_out/${1:-r04u}; mkdir -p $out; log=$out/conv_ph2.log
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_conv_direct.py tests/test_gpu_resnet.py tests/test_gpu_dpt.py tests/test_gpu_moco.py -m gpu -q -x -p no:cacheprovider > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $log
tail -2 $out/tests.log >> $log
for r in 1 2; do for p in 0 1; do
  for w in "--workload moco" "--workload depth --batch 128"; do
    echo "== PH2=$p $w" >> $log
    SSL4GIE_DEBUG_LIB=1 SSL4GIE_NT256_PH2=$p timeout -k 10 300 python bench.py $w --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $log
  done
done; done
cat $log
