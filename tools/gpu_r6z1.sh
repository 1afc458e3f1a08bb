#!/bin/bash
# round 6 final evidence, part 1: whole GPU suite + smoke + every workload + the default bench line
set -u
bash tools/gpu_final.sh r06
timeout -k 10 500 python bench.py > gpurun_out/r06_bench_default.log 2>&1; tail -1 gpurun_out/r06_bench_default.log | cut -c1-600
