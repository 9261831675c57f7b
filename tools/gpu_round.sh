#!/bin/bash
# One GPU-box visit: parity tests, the default bench line, the round profile.  Usage: bash tools/gpu_round.sh r03
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 3000 gpurun_out/${TAG}_bench.json
if [ -z "$NO_PROFILE" ]; then timeout 1500 bash tools/profile_round.sh $TAG; fi
