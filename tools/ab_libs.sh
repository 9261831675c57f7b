#!/bin/bash
# A/B of two builds of the library on one box, interleaved, the driver's command line (3 windows of 20 steps each):
#   bash tools/ab_libs.sh <rounds> <lib A> <lib B> ["bench flags"]     ("" = the default library)
cd $GRAFT_REPO_ROOT
R=$1; A=$2; B=$3; FLAGS=$4
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    echo "round $i | lib ${L:-default} | $(USHER_AMD_LIB=$L timeout 600 python bench.py --cpu-queries 0 --steps 20 --warmup 5 --no-extra $FLAGS 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]/1e6,3), [round(x,3) for x in d["windows"]["ms_per_step"]], r["kernel_ms_alone"], r["ms_per_step_alone"])')"
  done
done
