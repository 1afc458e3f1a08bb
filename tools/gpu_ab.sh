#!/bin/bash
# A/B of GEMM variants inside one gpurun call: usage gpu_ab.sh "<ENV=val ...>" "<ENV=val ...>" [rounds]
set -u
mkdir -p gpurun_out
rounds=${3:-2}
for r in $(seq 1 $rounds); do
  for v in "$1" "$2"; do
    echo "== $v"
    env $v timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^NT\|total" | cut -c1-12,44-80
  done
done
