#!/bin/bash
# Round profile: kernel-trace stats and the PMC passes (HBM traffic, L2, instruction issue) of the default bench
# workload.  Counters are collected in their own runs (--kernel-trace + --pmc only), one block set per pass.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02
# Writes gpurun_out/<tag>_*; tools/summarize_profile.py turns them into profiles/<tag>_*.
TAG=${1:-r02}
EXTRA=${2:-}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --queries 16384 --steps 3 --warmup 1 --cpu-queries 0 --no-extra --repeats 1 $EXTRA"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o p --output-format csv -- $B > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o p --output-format csv -- $B > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o p --output-format csv -- $B > gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/${TAG}_tcc -o p --output-format csv -- $B > gpurun_out/${TAG}_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/${TAG}_sq -o p --output-format csv -- $B > gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/${TAG}_sq2 -o p --output-format csv -- $B > gpurun_out/${TAG}_sq2.log 2>&1
tail -1 gpurun_out/${TAG}_stats.log
