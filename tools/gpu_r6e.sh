#!/bin/bash
# round 6: SyncBatchNorm with the single-process fusions + momentum-branch overlap across ranks (2 ranks on the one GPU, gloo)
set -u
out=gpurun_out/r6e; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "(tests: see r6f)"
export SSL4GIE_DIST_BACKEND=gloo SSL4GIE_BENCH_SAME_DATA=1
line() { grep -a "^{" | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('final_loss'), d.get('host_enqueue_ms'), {k: v for k, v in d.get('dp', {}).items() if 'syncbn' in k or k == 'transport'}, {k: (v['launches_per_step'], v['ms_per_step']) for k, v in d.get('roofline', {}).get('kernels', {}).items()})"; }
A="--workload moco --steps 3 --warmup 2 --batch 32 --no-cpu-baseline"
echo "== moco 1 rank" | tee -a $out/moco.log; timeout -k 10 300 python bench.py --gpus 1 $A 2>/dev/null | line | tee -a $out/moco.log
for v in "SSL4GIE_SYNCBN_FUSED=1 SSL4GIE_MOCO_OVERLAP_RANKS=1" "SSL4GIE_SYNCBN_FUSED=0 SSL4GIE_MOCO_OVERLAP_RANKS=0" "SSL4GIE_SYNCBN_FUSED=1 SSL4GIE_MOCO_OVERLAP_RANKS=1 SSL4GIE_SYNCBN=direct"; do
  echo "== moco 2 ranks | $v" | tee -a $out/moco.log
  env $v timeout -k 10 400 python bench.py --gpus 2 $A 2>&1 | line | tee -a $out/moco.log
done
# kernel lists: 1 rank vs rank 0 of 2 ranks (rocprofv3 around each rank's own python; no launcher in between)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof1 -o p -- python3 $R/bench.py --gpus 1 $A --prof-steps 0 > $R/$out/prof1.log 2>&1
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
for v in 1 0; do
  export SSL4GIE_SYNCBN_FUSED=$v SSL4GIE_MOCO_OVERLAP_RANKS=$v
  (RANK=1 LOCAL_RANK=1 timeout -k 10 300 python3 $R/bench.py --gpus 2 $A --prof-steps 0 > $R/$out/prof2_f${v}_rank1.log 2>&1 &)
  RANK=0 LOCAL_RANK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof2_f$v -o p -- python3 $R/bench.py --gpus 2 $A --prof-steps 0 > $R/$out/prof2_f$v.log 2>&1
  sleep 3
done
cd $R
python tools/kernel_list_diff.py $out/prof1 $out/prof2_f1 lars_apply_kernel > $out/kernel_diff_1rank_vs_2rank_fused.txt 2>&1
python tools/kernel_list_diff.py $out/prof1 $out/prof2_f0 lars_apply_kernel > $out/kernel_diff_1rank_vs_2rank_unfused.txt 2>&1
rm -rf $out/prof1 $out/prof2_f1 $out/prof2_f0
head -50 $out/kernel_diff_1rank_vs_2rank_fused.txt
