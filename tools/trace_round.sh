#!/bin/bash
# kernel trace of a 20-step pipelined run (third bound pinned on) for chain_gaps.py / pipeline_timeline.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp UGP_BOUND3=1
rocprofv3 --kernel-trace -d gpurun_out/r05_trace -o p --output-format csv -- python3 bench.py --cpu-queries 0 --steps 20 --warmup 5 --no-extra --repeats 1 > gpurun_out/r05_trace.log 2>&1
python3 tools/analysis/chain_gaps.py $(ls gpurun_out/r05_trace/*kernel_trace.csv gpurun_out/r05_trace/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/r05_chain_gaps.txt 2>&1
python3 tools/analysis/pipeline_timeline.py $(ls gpurun_out/r05_trace/*kernel_trace.csv gpurun_out/r05_trace/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/r05_pipeline_timeline.txt 2>&1
rm -rf gpurun_out/r05_trace
tail -5 gpurun_out/r05_chain_gaps.txt; tail -12 gpurun_out/r05_pipeline_timeline.txt
