#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "$1 $2: $(env $1 timeout 120 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
export UGP_SPLIT_CYCLES=400000
for w in 6 9 12 15 17; do run "UGP_WAVES_PER_CU=$w" ""; done
for s in 5 6 7 8 11 13; do run "UGP_LDS_SLOTS=$s" ""; done
run "UGP_LDS_SLOTS=7 UGP_WAVES_PER_CU=17" ""
run "UGP_REFILL_ALL=1" ""
run "UGP_CHUNK_NODES=150" ""
run "UGP_CHUNK_NODES=600" ""
