#!/bin/bash
# the non-headline BASELINE configurations (parity is checked in tests; these are timing notes for DESIGN.md)
run() { echo "$1 | $2 | $(env $1 timeout 600 python bench.py --cpu-queries 0 --steps 3 $2 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"])')"; }
run A=1 "--ambiguous"
run A=1 "--nodes 100000 --queries 1024"
run UGP_COARSE_MIN_NODES=0 "--nodes 100000 --queries 1024"
run A=1 "--nodes 100000 --queries 16384"
run UGP_COARSE_MIN_NODES=0 "--nodes 100000 --queries 16384"
run A=1 "--nodes 1000000 --queries 16384"
run A=1 "--queries 65536"
run A=1 "--queries 262144"
