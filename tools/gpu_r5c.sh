#!/bin/bash
# round 5: k-split TN kernel — exact-integer tests, then the pair bench old vs new in one box
set -u
out=gpurun_out/r5c; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_production_shapes.py -m gpu -q -x -k "tn or weight or wgrad or pair or group" --timeout 500 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for k in 0 1 0 1; do
  echo "== SSL4GIE_TN256K=$k"
  SSL4GIE_TN256K=$k TN_MODES=cold timeout -k 10 200 python tools/tn_pair_bench.py 2>&1 | grep TN-pair | sed 's/(GEMM + 2 slab reductions)//' | tee -a $out/pair_k$k.log
done
for k in 0 1; do
  echo "== bench SSL4GIE_TN256K=$k"
  SSL4GIE_TN256K=$k timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_k$k.log 2>&1; python - <<PY
import json
l=[x for x in open("$out/bench_k$k.log") if x.startswith("{")][-1]; d=json.loads(l)
print("ms_per_step", d["ms_per_step"], "img/s", d["value"], {k:(v["ms_per_step"], v["tflops"]) for k,v in d["roofline"]["kernels"].items()})
PY
done
