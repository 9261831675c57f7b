#!/bin/bash
# the non-headline BASELINE configurations (parity is checked in tests; these are timing notes for DESIGN.md):
# value, ms per step (pipelined), k_best8 ms in the timed region / alone, one step alone
run() { echo "$1 | $2 | $(env $1 timeout 900 python bench.py --cpu-queries 0 --steps ${STEPS:-10} --warmup 3 --no-extra --repeats 1 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_alone"], r["ms_per_step_alone"], d["host_buffer_path"]["placements_per_s"])')"; }
run A=1 ""
run A=1 "--no-overlap"
run A=1 "--ambiguous"
run A=1 "--ambiguous --iupac-true"
run UGP_NMASK=0 "--ambiguous"
run A=1 "--nodes 100000 --queries 1024"
run UGP_COARSE_MIN_NODES=0 "--nodes 100000 --queries 1024"
run A=1 "--nodes 100000 --queries 16384"
run A=1 "--nodes 1000000 --queries 16384"
run A=1 "--queries 10000"
run A=1 "--queries 65536"
STEPS=4 run A=1 "--queries 262144"
run A=1 "--shape sars2"
run A=1 "--shape sars2 --queries 65536"
