#!/bin/bash
# rocprofv3 kernel-trace summary of the Fitch-Sankoff bench (run on the GPU box through gpurun).
N=${1:-10000000}; S=${2:-2048}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fitch_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/fitch_stats -o p --output-format csv -- python3 tools/bench_fitch.py --nodes $N --sites $S --check-sites 0 --reps 1 > gpurun_out/fitch_stats.log 2>&1
grep '"metric"' gpurun_out/fitch_stats.log
f=$(find gpurun_out/fitch_stats -name "*kernel_stats.csv" | head -1)
grep -E "Name|k_fs|Radix|radix" $f | cut -c1-220
