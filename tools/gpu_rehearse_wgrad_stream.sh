#!/bin/bash
export SSL4GIE_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
run2() { timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 "${@:2}"; }
for v in 1 0; do
  export SSL4GIE_CONV_WGRAD_STREAM=$v
  echo "== depth 2 ranks, single-layer wgrad stream=$v"; run2 2955$v --workload depth --steps 3 --warmup 1 --batch 16 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" | cut -c1-1500
  echo "== moco 2 ranks (SyncBN), stream=$v"; run2 2956$v --workload moco --steps 3 --warmup 1 --batch 32 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" | cut -c1-1500
done
