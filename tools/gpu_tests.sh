#!/bin/bash
# Runs on the GPU box (via gpurun): per-op parity, model parity; stops after a timeout/kill.
set -u
mkdir -p gpurun_out
run() { # name, timeout_s, command...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.txt
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.txt
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL in $name: stopping" | tee -a gpurun_out/summary.txt; exit $rc; fi
  return $rc
}
: > gpurun_out/summary.txt
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" >> gpurun_out/summary.txt
for what in "$@"; do
  case $what in
    ops)   run ops 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 600 -p no:cacheprovider ;;
    opsall) run opsall 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider ;;
    mae)   run mae 1100 python -m pytest tests/test_gpu_mae.py -m gpu -q --timeout 900 -p no:cacheprovider ;;
    all)   run all 1100 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider ;;
    smoke) run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 900 python bench.py --steps 10 --warmup 3 ;;
    benchq) run benchq 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ;;
  esac
done
echo "=== done" >> gpurun_out/summary.txt
