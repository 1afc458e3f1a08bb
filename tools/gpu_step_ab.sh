#!/bin/bash
# A/B of whole-step variants inside one box: usage gpu_step_ab.sh "<ENV=val ...>" "<ENV=val ...>" [workload args]
set -u
mkdir -p gpurun_out
a=$1; b=$2; shift 2
for r in 1 2; do
  for v in "$a" "$b"; do
    echo "== $v"
    env $v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('final_loss'))"
  done
done
