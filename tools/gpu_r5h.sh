#!/bin/bash
set -u
out=gpurun_out/r5h; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -s > $out/all_tests.log 2>&1; echo "all tests rc=$?"; tail -8 $out/all_tests.log; grep "G14\|G13\|G15" $out/all_tests.log | cut -c1-400
timeout -k 10 300 python tools/moco_tn_shapes.py > $out/moco_tn_shapes.log 2>&1; cat $out/moco_tn_shapes.log | tail -60
