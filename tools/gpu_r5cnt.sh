#!/bin/bash
set -u
out=gpurun_out/r5cnt; mkdir -p $out
for w in moco depth; do for r in 1 2; do for lib in "" xcnt; do
  res=$(SSL4GIE_DEBUG_LIB=$lib timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d.get('final_loss'))")
  echo "$w round $r | lib=$lib | $res" | tee -a $out/sweep.log
done; done; done
