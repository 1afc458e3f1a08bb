#!/bin/bash
# A/B of MoCo-step variants inside one box: each argument is an "ENV=val ..." string
set -u
mkdir -p gpurun_out
for r in 1 2; do
  for v in "$@"; do
    echo "== $v"
    env $v timeout -k 10 300 python bench.py --workload moco --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('final_loss'))"
  done
done
