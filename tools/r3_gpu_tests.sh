#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3g_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r3g_pytest.log; tail -5 gpurun_out/r3g_pytest.log
timeout 600 python bench.py > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench.err; tail -c 1200 gpurun_out/r3g_bench.json
