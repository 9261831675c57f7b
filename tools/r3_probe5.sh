#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "random or knobs or locality or edge or polytom or caterpillar or config2 or global or syn or sixteen or sub_batch" > gpurun_out/r3e_pytest.log 2>&1; tail -3 gpurun_out/r3e_pytest.log
run() { echo "$1 $2: $(env $1 timeout 300 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
run A=1 ""
run UGP_SPLIT_CYCLES=0 ""
run "UGP_UNIT_GROW=0 UGP_SPLIT_CYCLES=0" ""
run "UGP_UNIT_GROW=0" ""
for sc in 50000 100000 400000; do run "UGP_SPLIT_CYCLES=$sc" ""; done
run "UGP_UNIT_GROW=2 UGP_UNIT_MAX=1024" ""
run "UGP_UNIT_GROW=2 UGP_UNIT_MAX=1024 UGP_SPLIT_CYCLES=100000" ""
run "UGP_UNIT_GROW=1 UGP_UNIT_MAX=4096 UGP_SPLIT_CYCLES=100000" ""
run "UGP_HEAVY_CHUNKS=64 UGP_SPLIT_CYCLES=100000" ""
run A=1 "--shape sars2"
run A=1 "--queries 65536"
UGP_STATS=1 timeout 600 python bench.py --cpu-queries 0 --steps 2 > gpurun_out/r3e_stats.json 2> gpurun_out/r3e_stats.err; grep "ugp stats" gpurun_out/r3e_stats.err | tail -9
