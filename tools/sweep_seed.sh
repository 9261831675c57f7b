#!/bin/bash
# sweeps with seeded upper bounds: unit size, chunk size
run() { echo "$1 $(env $1 timeout 300 python bench.py --cpu-queries 0 --steps 3 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["merge_ms"])')"; }
for u in 1 2 8 16; do run UGP_UNIT_CHUNKS=$u; done
for c in 600 1200 5000 10000; do run UGP_CHUNK_NODES=$c; done
run "UGP_CHUNK_NODES=10000 UGP_UNIT_CHUNKS=1"
run "UGP_CHUNK_NODES=20000 UGP_UNIT_CHUNKS=1"
