#!/bin/bash
# Rehearsal of the multi-rank code path on a ONE-GPU box: 2 ranks share the GPU, collectives go
# through gloo (RCCL refuses two ranks on one device).  Exercises DataParallel's bucket hooks, comm
# stream, the in-backward close of a pass (no finish()), the SyncBatchNorm exchange (torch.distributed and
# the library's peer-to-peer one) and the Barlow Twins all-reduce with real HIP kernels; the last line is
# `python bench.py --gpus 2` invoked plainly (no torchrun): bench.py starts its own ranks.
# With SSL4GIE_BENCH_SAME_DATA=1 the 2-rank MAE loss must equal the 1-rank loss at the same batch.
set -u
mkdir -p gpurun_out
export SSL4GIE_DIST_BACKEND=gloo SSL4GIE_BENCH_SAME_DATA=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run2() { timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 "${@:2}"; }
echo "== mae 1 rank"; timeout -k 10 300 python bench.py --gpus 1 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | cut -c1-900
echo "== mae 2 ranks (same data)"; run2 29511 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" | cut -c1-900
echo "== mae 2 ranks (same data), direct all-reduce transport (csrc/allreduce.hip over HIP IPC on the one device)"; SSL4GIE_ALLREDUCE=direct run2 29515 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" | cut -c1-900
echo "== mae 2 ranks (same data), SSL4GIE_ALLREDUCE=auto: both transports probed on a 64-MiB slice, verdict in dp.transport_probe"; SSL4GIE_ALLREDUCE=auto run2 29517 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" | cut -c1-1100
echo "== depth 2 ranks"; run2 29512 --workload depth --steps 2 --warmup 1 --batch 16 2>&1 | grep -a "^{\|Error\|error" | cut -c1-600
echo "== moco 2 ranks (SyncBN)"; run2 29513 --workload moco --steps 2 --warmup 1 --batch 32 2>&1 | grep -a "^{\|Error\|error" | cut -c1-600
echo "== bt 2 ranks"; run2 29514 --workload bt --steps 2 --warmup 1 --batch 64 2>&1 | grep -a "^{\|Error\|error" | cut -c1-600
echo "== moco 2 ranks (SyncBN carried by the direct exchange: 0 torch.distributed collectives for it)"; SSL4GIE_SYNCBN=direct run2 29516 --workload moco --steps 2 --warmup 1 --batch 32 2>&1 | grep -a "^{\|Error\|error" | cut -c1-700
echo "== moco 2 ranks, SSL4GIE_SYNCBN=auto (probe in dp.syncbn_probe)"; SSL4GIE_SYNCBN=auto run2 29518 --workload moco --steps 2 --warmup 1 --batch 32 2>&1 | grep -a "^{\|Error\|error" | cut -c1-900
echo "== mae, plain 'python bench.py --gpus 2' (self-launched ranks)"; timeout -k 10 500 python bench.py --gpus 2 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --prof-steps 0 2>&1 | grep -a "^{" | cut -c1-900
