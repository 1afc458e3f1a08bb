#!/bin/bash
# round 5: new production-shape tests first, then the whole GPU suite, then every workload's bench line
set -u
out=gpurun_out/r5g; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_shapes_configs.py -m gpu -q --timeout 800 -p no:cacheprovider > $out/prod_tests.log 2>&1; echo "prod tests rc=$?"; tail -25 $out/prod_tests.log
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_production_shapes_configs.py > $out/all_tests.log 2>&1; echo "all tests rc=$?"; tail -5 $out/all_tests.log
for w in mae moco depth vit bt det; do
  timeout -k 10 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$w.log 2>&1; echo "== $w rc=$?"; tail -1 $out/bench_$w.log | cut -c1-1500
done
