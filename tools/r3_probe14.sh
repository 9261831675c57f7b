#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "$1 $2: $(env $1 timeout 200 python bench.py --cpu-queries 0 --steps 20 --warmup 3 $2 2>gpurun_out/err.txt | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_alone"], r["merge_ms"], r["coarse_ms"], r["table_ms"])')"; }
run A=1 ""
run UGP_KBEST_SHARED=1 ""
run A=1 "--shape sars2"
run A=1 "--queries 65536"
run A=1 "--ambiguous"
run A=1 "--nodes 100000 --queries 1024"
run UGP_WAVES_PER_CU=16 ""
run UGP_WAVES_PER_CU=15 ""
