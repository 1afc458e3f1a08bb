#!/bin/bash
# model-level parity with the prefetching attention backward + vit / depth / bt bench A/B
set -u
out=gpurun_out/r5zb; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "attention or vit or depth or dpt or barlow or bt or g13 or g15 or finetune or det or block" > $out/tests.log 2>&1; rc=$?; echo "model tests rc=$rc"; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit 1
for w in vit depth; do for k in 0 1; do res=$(SSL4GIE_ATTN_PREFETCH=$k python bench.py --workload $w --steps 12 --warmup 4 --prof-steps 0 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['value'], d.get('model_mfma_frac'), d.get('final_loss'))"); echo "$w prefetch=$k | $res" | tee -a $out/bench.log; done; done
