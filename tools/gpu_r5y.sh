#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r5y; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 bench.py --steps 4 --warmup 2 --prof-steps 0 --no-cpu-baseline > $out/run.log 2>&1
find $out/raw -name "*kernel_trace.csv" -exec cp {} $out/trace.csv \;
rm -rf $out/raw
python3 tools/copy_neighbours.py $out/trace.csv | tee $out/copy_neighbours.txt
rm -f $out/trace.csv
