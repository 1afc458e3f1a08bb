#!/bin/bash
# round 6 final evidence, part 3: two-rank rehearsal on the one GPU (gloo) incl. the transport probes
set -u
bash tools/gpu_rehearse_dp.sh > gpurun_out/r06_dp_rehearsal.log 2>&1; cut -c1-420 gpurun_out/r06_dp_rehearsal.log
