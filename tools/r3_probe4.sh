#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "random or knobs or locality or edge or polytom or caterpillar or config2 or global or syn or sixteen or sub_batch" > gpurun_out/r3d_pytest.log 2>&1; tail -3 gpurun_out/r3d_pytest.log
run() { echo "$1 $2: $(env $1 timeout 600 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
run A=1 ""
run UGP_UNIT_GROW=0 ""
run UGP_UNIT_GROW=2 ""
run UGP_UNIT_GROW=8 ""
run "UGP_UNIT_GROW=4 UGP_UNIT_MAX=64" ""
run "UGP_UNIT_GROW=4 UGP_UNIT_MAX=1024" ""
run "UGP_UNIT_GROW=2 UGP_UNIT_MAX=1024" ""
run "UGP_UNIT_GROW=4 UGP_HEAVY_CHUNKS=8" ""
run "UGP_UNIT_GROW=4 UGP_UNIT_CHUNKS=8" ""
run A=1 "--shape sars2"
run UGP_UNIT_GROW=0 "--shape sars2"
run A=1 "--queries 65536"
UGP_STATS=1 timeout 600 python bench.py --cpu-queries 0 --steps 2 > gpurun_out/r3d_stats.json 2> gpurun_out/r3d_stats.err; grep "ugp stats" gpurun_out/r3d_stats.err | tail -9
