#!/bin/bash
# round 5: parity tests, then the bench with the third pruning bound pinned on / left to the handle, on the three workloads; kernel table last
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bound3_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -3
run() { echo "$1 $2 | $(env $1 timeout 400 python bench.py --cpu-queries 8 --steps 20 --warmup 8 --no-extra --repeats 3 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["windows"]["ms_per_step"], "kernel", r["kernel_ms"], r["kernel_ms_in_region"], "alone", r["ms_per_step_alone"], "table", r["table_ms"], "merge", r["merge_ms"], "mism", d["cpu_baseline"]["mismatches_vs_gpu"], d["config"]["last_step_equals_stream_ordered_call"])')"; }
run "UGP_BOUND3=1" ""
run "X=1" ""
run "UGP_BOUND3=1" "--ambiguous"
run "X=1" "--shape sars2"
UGP_BOUND3=1 bash tools/kernel_table.sh r05u | grep -E "b3|k_ties|best8|k_descend"
