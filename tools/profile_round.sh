#!/bin/bash
# Round profile: kernel-trace stats and HBM-traffic PMC of the default bench workload.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
# Writes gpurun_out/<tag>_*; tools/summarize_profile.py turns them into profiles/<tag>_*.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --queries 16384 --steps 3 --warmup 1 --cpu-queries 0"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o p --output-format csv -- $B > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o p --output-format csv -- $B > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o p --output-format csv -- $B > gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/${TAG}_tcc -o p --output-format csv -- $B > gpurun_out/${TAG}_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/${TAG}_sq -o p --output-format csv -- $B > gpurun_out/${TAG}_sq.log 2>&1
tail -1 gpurun_out/${TAG}_stats.log
