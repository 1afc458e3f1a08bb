#!/bin/bash
set -u
out=gpurun_out/r6b; mkdir -p $out
timeout -k 10 300 python tools/check_gelu_table.py > $out/check.log 2>&1
