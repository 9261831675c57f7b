#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "A=1" "UGP_SPLIT_CYCLES=250000" "UGP_UNIT_GROW=0 UGP_SPLIT_CYCLES=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg UGP_STATS=1 UGP_TRACE=gpurun_out/trace_$tag.bin timeout 300 python bench.py --cpu-queries 0 --steps 1 --warmup 1 > gpurun_out/trace_$tag.json 2> gpurun_out/trace_$tag.err
  echo "== $cfg: $(python -c "import json;d=json.load(open('gpurun_out/trace_$tag.json'));print(d['roofline']['kernel_ms'])")"
  python tools/analysis/unit_trace.py gpurun_out/trace_$tag.bin
done
