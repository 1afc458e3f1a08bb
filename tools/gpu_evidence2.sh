#!/bin/bash
# round-2 extras: rehearsal incl. the direct transport; TN traffic with grouped weight gradients
set -u
mkdir -p gpurun_out
bash tools/gpu_rehearse_dp.sh > gpurun_out/r02f_rehearse.log 2>&1
cut -c1-400 gpurun_out/r02f_rehearse.log
SSL4GIE_WGRAD_GROUP=4 SSL4GIE_WGRAD_STREAM=0 bash tools/gpu_pmc.sh r02f_group bench.py --steps 2 --warmup 1 --prof-steps 0 --no-cpu-baseline > gpurun_out/r02f_pmc_group.txt 2>&1
grep -a "gemm_bf16_tn256\|slab_reduce" gpurun_out/pmc_r02f_group/pmc_summary.txt | cut -c1-260
