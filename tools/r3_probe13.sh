#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "consecutive or sub_batch or edge" 2>&1 | tail -2
for a in "" "--steps 20 --warmup 5" "--steps 5 --warmup 0"; do
timeout 600 python bench.py --cpu-queries 0 $a 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["value"], d["ms_per_step"], d["steps"], r["frac"], r["kernel_ms"], r["kernel_ms_alone"], r["frac_alone"], r["ms_per_step_alone"])'
done
