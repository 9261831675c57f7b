#!/bin/bash
# round-5 experiment: walks that take turns (UGP_KBEST_EXCLUSIVE, exp library) on most of the wave slots, helpers beside them
cd $GRAFT_REPO_ROOT
run() { echo "$1 | $(env $1 timeout 300 python bench.py --cpu-queries 0 --steps 20 --warmup 5 --no-extra --repeats 3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["windows"]["ms_per_step"], r["kernel_ms"], r["kernel_ms_in_region"], r["ms_per_step_alone"], d["config"]["last_step_equals_stream_ordered_call"])')"; }
run "BENCH_EXP_LIB=1"
run "BENCH_EXP_LIB=1 UGP_KBEST_EXCLUSIVE=1 UGP_WAVES_PER_CU=12"
run "BENCH_EXP_LIB=1 UGP_KBEST_EXCLUSIVE=1 UGP_WAVES_PER_CU=14"
run "BENCH_EXP_LIB=1 UGP_KBEST_EXCLUSIVE=1 UGP_WAVES_PER_CU=10"
run "BENCH_EXP_LIB=1 UGP_KBEST_EXCLUSIVE=1 UGP_WAVES_PER_CU=12 UGP_PIPELINE_DEPTH=4"
run "BENCH_EXP_LIB=1 UGP_KBEST_EXCLUSIVE=1 UGP_WAVES_PER_CU=12 UGP_PIPELINE_DEPTH=2"
run "BENCH_EXP_LIB=1 UGP_WAVES_PER_CU=8"
