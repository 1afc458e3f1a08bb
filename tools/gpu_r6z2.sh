#!/bin/bash
# round 6 final evidence, part 2: rocprofv3 kernel stats (mae, depth, moco) + PMC passes of the bench line
set -u
bash tools/gpu_prof.sh r06 > /dev/null 2>&1; head -12 gpurun_out/prof_r06/kernel_stats.csv | cut -c1-200
bash tools/gpu_prof.sh r06_depth --workload depth --batch 128 > /dev/null 2>&1
bash tools/gpu_prof.sh r06_moco --workload moco > /dev/null 2>&1
bash tools/gpu_pmc.sh r06 bench.py --steps 3 --warmup 2 --prof-steps 0 --no-cpu-baseline > gpurun_out/pmc_r06.log 2>&1; tail -30 gpurun_out/pmc_r06.log | cut -c1-250
