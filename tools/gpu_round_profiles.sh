#!/bin/bash
# Round-end evidence on one box: bench line (with roofline + cpu_baseline), rocprofv3 kernel stats of the
# MAE / depth / MoCo steps, the three PMC passes of the MAE step, all workloads back to back.
set -u
tag=${1:-r02}
mkdir -p gpurun_out
echo "== bench (default flags)"; timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log | cut -c1-400
echo "== kernel stats mae"; bash tools/gpu_prof.sh ${tag}_mae > gpurun_out/${tag}_prof_mae.txt 2>&1; tail -3 gpurun_out/${tag}_prof_mae.txt | cut -c1-200
echo "== kernel stats depth"; bash tools/gpu_prof.sh ${tag}_depth --workload depth --batch 128 > gpurun_out/${tag}_prof_depth.txt 2>&1
echo "== kernel stats moco"; bash tools/gpu_prof.sh ${tag}_moco --workload moco > gpurun_out/${tag}_prof_moco.txt 2>&1
echo "== pmc mae"; SSL4GIE_WGRAD_STREAM=0 bash tools/gpu_pmc.sh ${tag}_mae bench.py --steps 2 --warmup 1 --prof-steps 0 --no-cpu-baseline > gpurun_out/${tag}_pmc.txt 2>&1; tail -5 gpurun_out/${tag}_pmc.txt | cut -c1-300
echo "== all workloads"
for w in mae vit depth moco bt det; do
  extra=""; [ $w = depth ] && extra="--batch 128"; [ $w = bt ] && extra="--batch 512"; [ $w = det ] && extra="--batch 4"
  timeout -k 10 400 python bench.py --workload $w $extra --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>&1 | tail -1 | cut -c1-700
done > gpurun_out/${tag}_all_workloads.log 2>&1
cat gpurun_out/${tag}_all_workloads.log | cut -c1-300
