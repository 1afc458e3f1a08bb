#!/bin/bash
set -u
out=gpurun_out/r5i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_curves.py tests/test_gpu_moco.py tests/test_gpu_resnet.py tests/test_gpu_production_shapes_configs.py tests/test_gpu_ops.py -m gpu -q --timeout 800 -p no:cacheprovider -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -6 $out/tests.log; grep "^G14\|G14 " $out/tests.log | cut -c1-600
timeout -k 10 300 python tools/moco_tn_shapes.py 2>&1 | grep OLD
for w in moco mae; do
  timeout -k 10 400 python bench.py --workload $w --steps 15 --warmup 4 --no-cpu-baseline > $out/bench_$w.log 2>&1; echo "== $w rc=$?"; tail -1 $out/bench_$w.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['value'], {k:(v['ms_per_step'], v.get('tflops', v.get('GBs'))) for k,v in d['roofline']['kernels'].items()})"
done
