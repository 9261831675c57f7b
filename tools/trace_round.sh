#!/bin/bash
# kernel trace of a 20-step pipelined run for chain_gaps.py / pipeline_timeline.py:   bash tools/trace_round.sh r06 ["bench flags"]
TAG=${1:-r06}
EXTRA=${2:-}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_trace -o p --output-format csv -- python3 bench.py --cpu-queries 0 --steps 20 --warmup 5 --no-extra --repeats 1 $EXTRA > gpurun_out/${TAG}_trace.log 2>&1
python3 tools/analysis/chain_gaps.py $(ls gpurun_out/${TAG}_trace/*kernel_trace.csv gpurun_out/${TAG}_trace/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/${TAG}_chain_gaps.txt 2>&1
python3 tools/analysis/pipeline_timeline.py $(ls gpurun_out/${TAG}_trace/*kernel_trace.csv gpurun_out/${TAG}_trace/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/${TAG}_pipeline_timeline.txt 2>&1
rm -rf gpurun_out/${TAG}_trace
tail -5 gpurun_out/${TAG}_chain_gaps.txt; tail -12 gpurun_out/${TAG}_pipeline_timeline.txt
