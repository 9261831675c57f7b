#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "$1 $2: $(env $1 timeout 120 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
run UGP_LIGHT_ORDER=1 ""
run UGP_LIGHT_ORDER=0 ""
run "UGP_LIGHT_ORDER=1 UGP_SPLIT_CYCLES=250000" ""
run "UGP_LIGHT_ORDER=1 UGP_SPLIT_CYCLES=800000" ""
run UGP_LIGHT_ORDER=0 "--shape sars2"
run UGP_LIGHT_ORDER=1 "--shape sars2"
run "UGP_UB_EVERY=32" ""
run "UGP_UB_EVERY=1000" ""
run "UGP_PRUNE_MIN_WORDS=8" ""
run "UGP_PRUNE_MIN_WORDS=16" ""
run "UGP_PRUNE_MIN_WORDS=2" ""
