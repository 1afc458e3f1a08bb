#!/bin/bash
set -u
export SSL4GIE_DIST_BACKEND=gloo SSL4GIE_BENCH_SAME_DATA=1 HSA_ENABLE_IPC_MODE_LEGACY=0
SSL4GIE_ALLREDUCE=auto timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" > gpurun_out/r06_mae_2rank_auto.json
python -c "
import json; d=json.loads(open('gpurun_out/r06_mae_2rank_auto.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'], json.dumps(d['dp']))"
