#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "random or knobs or locality or edge or polytom or caterpillar or config2 or global or syn or sixteen" > gpurun_out/r3c_pytest.log 2>&1; tail -3 gpurun_out/r3c_pytest.log
run() { echo "$1 $2: $(env $1 timeout 600 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
run A=1 ""
for sj in 0 16 32 128 512; do run UGP_SHORT_JUMP=$sj ""; done
run A=1 "--shape sars2"
run UGP_SHORT_JUMP=0 "--shape sars2"
UGP_STATS=1 timeout 600 python bench.py --cpu-queries 0 --steps 2 > gpurun_out/r3c_stats.json 2> gpurun_out/r3c_stats.err; grep "ugp stats" gpurun_out/r3c_stats.err | tail -9
