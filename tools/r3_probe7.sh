#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "$1 $2: $(env $1 timeout 120 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
for sc in 250000 400000 600000 1000000 2000000; do run "UGP_SPLIT_CYCLES=$sc" ""; done
for sc in 400000 1000000; do
run "UGP_SPLIT_CYCLES=$sc UGP_UNIT_GROW=8" ""
run "UGP_SPLIT_CYCLES=$sc UGP_UNIT_GROW=2" ""
run "UGP_SPLIT_CYCLES=$sc UGP_UNIT_MAX=64" ""
run "UGP_SPLIT_CYCLES=$sc UGP_UNIT_MAX=1024" ""
run "UGP_SPLIT_CYCLES=$sc UGP_HEAVY_CHUNKS=32" ""
run "UGP_SPLIT_CYCLES=$sc UGP_HEAVY_CHUNKS=64" ""
run "UGP_SPLIT_CYCLES=$sc UGP_UNIT_GROW=0" ""
run "UGP_SPLIT_CYCLES=$sc" "--shape sars2"
run "UGP_SPLIT_CYCLES=$sc" "--queries 65536"
done
