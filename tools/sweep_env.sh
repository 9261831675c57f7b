#!/bin/bash
# one bench line per environment setting:  bash tools/sweep_env.sh "bench flags" "A=1" "B=2 C=3" ...   (value M/s, ms/step, kernel alone, step alone, mismatches)
cd $GRAFT_REPO_ROOT
FLAGS=$1; shift
for setting in "$@"; do
  echo "$setting | $(env $setting timeout 600 python bench.py --cpu-queries 4 --steps 12 --warmup 6 --no-extra --repeats 1 $FLAGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]/1e6,3), d["ms_per_step"], r["kernel_ms_alone"], r["ms_per_step_alone"], r["coarse_ms"], r["table_ms"], r["merge_ms"], (d.get("cpu_baseline") or {}).get("mismatches_vs_gpu"))')"
done
