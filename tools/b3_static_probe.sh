#!/bin/bash
# The third pruning bound on / off for every (tree shape x query kind) the static rule has to decide (ugp_tuner.hpp b3_static_choice):
# value (M placements/s, pipelined), k_best8 ms alone, one step alone.   bash tools/b3_static_probe.sh > gpurun_out/b3_probe.txt
cd $GRAFT_REPO_ROOT
run() { for b in 0 1; do export UGP_BOUND3=$b; echo "$1 | UGP_BOUND3=$b | $(timeout 600 python bench.py --cpu-queries 0 --steps 12 --warmup 6 --no-extra --repeats 1 $1 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]/1e6,3), d["ms_per_step"], r["kernel_ms_alone"], r["ms_per_step_alone"], r["third_bound_steps"])')"; done; }
run ""
run "--ambiguous"
run "--shape sars2"
run "--shape sars2 --ambiguous"
run "--shape sars2 --nodes 15000000 --queries 10000"
run "--nodes 1000000"
run "--shape sars2 --nodes 1000000"
