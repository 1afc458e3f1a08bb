#!/bin/bash
# non-temporal epilogue stores chosen per launch (SSL4GIE_NT_STREAM_M: -1 never, 32768 default, 0 always)
set -u
out=gpurun_out/r5nt2; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_production_shapes_configs.py tests/test_gpu_ops.py -m gpu -q -x --timeout 800 -p no:cacheprovider > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit 1
for w in mae vit; do
  for r in 1 2; do for m in -1 32768 0; do
    res=$(SSL4GIE_NT_STREAM_M=$m timeout -k 10 300 python bench.py --workload $w --steps 12 --warmup 4 --no-cpu-baseline --prof-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d.get('final_loss'))")
    echo "$w round $r | NT_STREAM_M=$m | $res" | tee -a $out/sweep.log
  done; done
done
for m in -1 32768 0; do echo "== SSL4GIE_NT_STREAM_M=$m"; SSL4GIE_NT_STREAM_M=$m GEMM_SKIP_TN=1 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu" | tee $out/nt_m$m.log; done
