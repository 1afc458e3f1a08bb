#!/bin/bash
# GPU busy / idle accounting of the MoCo-R50 and MAE steps (kernel trace; union of kernel intervals)
set -u
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5x; mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in moco mae; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/raw_$w -- python3 bench.py --workload $w --steps 24 --warmup 3 --prof-steps 0 --no-cpu-baseline > $out/run_$w.log 2>&1
  echo "$w rc=$?"; tail -1 $out/run_$w.log | cut -c1-200
  find $out/raw_$w -name "*kernel_trace.csv" -exec cp {} $out/trace_$w.csv \;
  rm -rf $out/raw_$w
  python3 tools/trace_gaps.py $out/trace_$w.csv | tee $out/gaps_$w.txt
  rm -f $out/trace_$w.csv
done
