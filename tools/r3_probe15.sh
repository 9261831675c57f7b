#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "random or knobs or locality or edge or polytom or caterpillar or config2 or global or syn or sixteen or sub_batch or full_size or consecutive" 2>&1 | tail -2
run() { echo "$1 $2: $(env $1 timeout 200 python bench.py --cpu-queries 0 --steps 20 --warmup 3 $2 2>gpurun_out/err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_alone"], r["ms_per_step_alone"])')"; }
run A=1 ""
run UGP_NO_LDS_BITS=1 ""
run UGP_LDS_SLOTS=7 ""
run UGP_LDS_SLOTS=6 ""
run UGP_LDS_SLOTS=11 ""
run A=1 "--shape sars2"
run UGP_NO_LDS_BITS=1 "--shape sars2"
run A=1 "--queries 65536"
run UGP_NO_LDS_BITS=1 "--queries 65536"
run A=1 "--ambiguous"
run UGP_NO_LDS_BITS=1 "--ambiguous"
