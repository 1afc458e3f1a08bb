#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> kernel stats + GPU busy / idle-gap accounting (tools/trace_gaps.py)
set -u
tag=${1:-gaps}; shift || true
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/gaps_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 bench.py --steps 6 --warmup 2 --prof-steps 0 --no-cpu-baseline "$@" > $out/bench.log 2>&1
echo "rc=$?" >> $out/bench.log
find $out/raw -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
tr=$(find $out/raw -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$tr" > $out/gaps.txt 2>&1
rm -rf $out/raw
cat $out/gaps.txt
