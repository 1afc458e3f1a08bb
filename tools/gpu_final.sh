#!/bin/bash
# final check of a tree on one box: whole GPU suite + smoke, then every bench workload back to back
set -u
tag=${1:-final}
mkdir -p gpurun_out
bash tools/gpu_tests.sh all smoke > gpurun_out/${tag}_tests.log 2>&1
# gpu_tests.sh runs pytest with -x and keeps going to the smoke step: its own exit code says nothing
if grep -aq "^rc=[1-9]\| failed\|FAILED" gpurun_out/${tag}_tests.log; then echo "TESTS FAILED"; else echo "tests ok"; fi
grep -a "passed\|failed\|FAILED\|smoke" gpurun_out/${tag}_tests.log | tail -6
for w in mae vit depth moco bt det; do
  extra=""; [ $w = depth ] && extra="--batch 128"; [ $w = bt ] && extra="--batch 512"; [ $w = det ] && extra="--batch 4"
  timeout -k 10 400 python bench.py --workload $w $extra --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>&1 | tail -1 | cut -c1-700
done > gpurun_out/${tag}_all_workloads.log 2>&1
cut -c1-60,100-260 gpurun_out/${tag}_all_workloads.log
