#!/bin/bash
# kernel time table (rocprofv3 --kernel-trace --stats) of one short bench run (args: tag, then bench flags)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o p --output-format csv -- python3 bench.py --cpu-queries 0 --steps 3 --warmup 1 --no-extra "$@" > /dev/null 2>&1
python3 - $TAG <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s_stats/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
for r in list(csv.DictReader(open(f[0])))[:22]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
