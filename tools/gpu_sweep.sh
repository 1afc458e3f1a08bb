#!/bin/bash
# whole-step sweep inside one box: usage gpu_sweep.sh <tag> "<ENV=val ...>" "<ENV=val ...>" ... ; two rounds each,
# bench.py --steps 15 --warmup 4 without the profiling pass; prints ms/step per variant
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for v in "$@"; do
    res=$(env $v timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --prof-steps 0 ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('final_loss'))")
    echo "round $r | $v | $res" | tee -a $out/sweep.log
  done
done
