#!/bin/bash
# sweep of chunk / unit sizes and the bound-exchange period; every line carries the oracle mismatch count
run() { echo "$1 | $(env $1 timeout 600 python bench.py --cpu-queries 6 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"], "mismatches", d["cpu_baseline"]["mismatches_vs_gpu"])')"; }
for e in "$@"; do run "$e"; done
