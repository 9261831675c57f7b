#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
UGP_STATS=1 timeout 600 python bench.py --cpu-queries 0 --steps 2 > gpurun_out/r3b_stats.json 2> gpurun_out/r3b_stats.err; grep "ugp stats" gpurun_out/r3b_stats.err | tail -8
UGP_STATS=1 timeout 600 python bench.py --cpu-queries 0 --steps 2 --shape sars2 > gpurun_out/r3b_stats_s.json 2> gpurun_out/r3b_stats_s.err; grep "ugp stats" gpurun_out/r3b_stats_s.err | tail -8
bash tools/pmc_icache.sh 2>&1 | tail -20
