#!/bin/bash
set -u
out=gpurun_out/r6f; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_syncbn_direct.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" | cut -c1-1500 | tail -8 | tee $out/tests.log
bash tools/gpu_r6e.sh > $out/r6e.log 2>&1
