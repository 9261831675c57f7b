#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "random or knobs or locality or edge or polytom or caterpillar or config2 or global or syn or sixteen or sub_batch or full_size" > gpurun_out/r3h_pytest.log 2>&1; tail -3 gpurun_out/r3h_pytest.log
run() { echo "$1 $2: $(env $1 timeout 120 python bench.py --cpu-queries 0 --steps 5 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], d["roofline"]["coarse_ms"])')"; }
run A=1 ""
run A=1 ""
run UGP_SPLIT_CYCLES=250000 ""
run UGP_SPLIT_CYCLES=600000 ""
run UGP_UNIT_GROW=0 ""
run A=1 "--shape sars2"
run A=1 "--queries 65536"
run A=1 "--ambiguous"
