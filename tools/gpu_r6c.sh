#!/bin/bash
set -u
out=gpurun_out/r6c; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_models_golden.py tests/test_gpu_allreduce_direct.py "tests/test_gpu_resnet.py::test_optimizer_step_between_forward_and_backward_is_refused_only_for_its_own_parameters" tests/test_gpu_ops.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | cut -c1-600 > $out/tests3.log
