#!/bin/bash
set -u
out=gpurun_out/r5nt3; mkdir -p $out
for w in moco depth bt det; do
  extra=""; [ $w = bt ] && extra="--batch 512"; [ $w = det ] && extra="--batch 4"
  for r in 1 2; do for m in -1 32768; do
    res=$(SSL4GIE_NT_STREAM_M=$m timeout -k 10 300 python bench.py --workload $w $extra --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d.get('final_loss'))")
    echo "$w round $r | NT_STREAM_M=$m | $res" | tee -a $out/sweep.log
  done; done
done
