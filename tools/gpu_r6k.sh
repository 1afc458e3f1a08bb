#!/bin/bash
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 800 python -m pytest tests/test_gpu_moco.py -q -m gpu -k "across_ranks" 2>&1 | grep -v amdgpu.ids | cut -c1-800 | tail -30 > gpurun_out/r6k_tests.log; tail -12 gpurun_out/r6k_tests.log
