#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "$1 $2: $(env $1 timeout 120 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
for l in 400000 1000000 3000000; do for h in 30000 60000 120000 250000; do run "UGP_SPLIT_CYCLES=$l UGP_SPLIT_HEAVY=$h" ""; done; done
run "UGP_SPLIT_CYCLES=1000000 UGP_SPLIT_HEAVY=60000 UGP_HEAVY_CHUNKS=64" ""
run "UGP_SPLIT_CYCLES=1000000 UGP_SPLIT_HEAVY=60000 UGP_HEAVY_CHUNKS=4" ""
run "UGP_SPLIT_CYCLES=1000000 UGP_SPLIT_HEAVY=60000" "--shape sars2"
run "UGP_SPLIT_CYCLES=1000000 UGP_SPLIT_HEAVY=60000" "--queries 65536"
