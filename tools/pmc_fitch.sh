#!/bin/bash
# counters of the Fitch-Sankoff kernels (own passes, --kernel-trace + --pmc only):  bash tools/pmc_fitch.sh [tag] ["bench_fitch flags"]   (GPU box, repo root)
# prints, for the widest levels of the two sweeps and for the whole call: HBM bytes read (FETCH_SIZE x 2, gfx950) / written, duration, TB/s,
# waves, the wait fraction
TAG=${1:-fitch}
EXTRA=${2:-}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 tools/bench_fitch.py --check-sites 0 --reps 1 $EXTRA"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  d=gpurun_out/${TAG}_pmc_$(echo $P | cut -d' ' -f1)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $P -d $d -o p --output-format csv -- $B > $d.log 2>&1
done
python3 tools/analysis/fitch_pmc.py $TAG
