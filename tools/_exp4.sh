#!/bin/bash
# round-5 experiment: third bound with the block-ordered event lists
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bound3_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -8
run() { echo "$1 $2 | $(env $1 timeout 400 python bench.py --cpu-queries 8 --steps 20 --warmup 5 --no-extra --repeats 3 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["windows"]["ms_per_step"], "kernel", r["kernel_ms"], r["kernel_ms_in_region"], "alone", r["ms_per_step_alone"], "table", r["table_ms"], "mism", d["cpu_baseline"]["mismatches_vs_gpu"], d["config"]["last_step_equals_stream_ordered_call"])')"; }
run "UGP_NO_BOUND3=1" ""
run "X=1" ""
run "UGP_NO_BOUND3=1" "--ambiguous"
run "X=1" "--ambiguous"
run "UGP_NO_BOUND3=1" "--shape sars2"
run "X=1" "--shape sars2"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05b3b_stats -o b3 -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-queries 0 --steps 3 --warmup 1 --no-extra > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05b3b_stats/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
