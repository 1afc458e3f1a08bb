#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r06_tests_full.log 2>&1; echo "rc=$?" >> gpurun_out/r06_tests_full.log
tail -8 gpurun_out/r06_tests_full.log | cut -c1-300
