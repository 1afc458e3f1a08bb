#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run; summaries copied to gpurun_out/prof_<tag>
set -u
tag=${1:-r01}; shift || true
# per-kernel durations are taken with every launch alone on the chip (as bench.py's HIP-event pass
# does): the block executor's weight-gradient side stream is folded back unless WGRAD_STREAM=1
export SSL4GIE_WGRAD_STREAM=${WGRAD_STREAM:-0}
# ... and so is the single layers' weight-gradient stream of the ResNet / DPT paths (engine.wgrad_fork)
export SSL4GIE_CONV_WGRAD_STREAM=${WGRAD_STREAM:-0}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 bench.py --steps 5 --warmup 2 --prof-steps 0 --no-cpu-baseline "$@" > $out/bench.log 2>&1
echo "rc=$?" >> $out/bench.log
find $out/raw -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out/raw -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > '$out'/kernel_trace_head.csv; wc -l "$1" >> '$out'/kernel_trace_head.csv' _ {} \;
rm -rf $out/raw
tail -3 $out/bench.log
head -40 $out/kernel_stats.csv
