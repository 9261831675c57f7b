#!/bin/bash
# interleaved A/B of environment settings, the driver's command line:  bash tools/ab_env.sh <rounds> "<bench flags>" "<ENV..>" "<ENV..>" ...
cd $GRAFT_REPO_ROOT
R=$1; FLAGS=$2; shift; shift
for i in $(seq 1 $R); do
  for setting in "$@"; do
    echo "round $i | $FLAGS | $setting | $(env $setting timeout 600 python bench.py --cpu-queries 0 --steps 20 --warmup 5 --no-extra $FLAGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,3), [round(x,3) for x in d["windows"]["ms_per_step"]])')"
  done
done
