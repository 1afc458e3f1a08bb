#!/bin/bash
set -u
for w in vit depth moco bt det; do
  extra=""; [ $w = depth ] && extra="--batch 128"; [ $w = bt ] && extra="--batch 512"; [ $w = det ] && extra="--batch 4"
  timeout -k 10 400 python bench.py --workload $w $extra --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
done > gpurun_out/r06_all_workloads_roofline.log 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r06_all_workloads_roofline.log'):
    if not line.startswith('{'): continue
    d = json.loads(line); r = d.get('roofline') or {}
    print(d['metric'][:40], d['ms_per_step'], d.get('model_mfma_frac'), r.get('kernel'), r.get('frac'), (r.get('hbm_bound_kernel') or {}).get('kernel'), (r.get('hbm_bound_kernel') or {}).get('frac'))
PY
