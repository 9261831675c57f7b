#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 bench.py --queries 16384 --steps 3 --warmup 1 --cpu-queries 0"
rm -rf gpurun_out/ic1 gpurun_out/ic2
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d gpurun_out/ic1 -o p --output-format csv -- $B > gpurun_out/ic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/ic2 -o p --output-format csv -- $B > gpurun_out/ic2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("ic1","ic2"):
    f = glob.glob("gpurun_out/%s/*counter_collection.csv" % d)
    if not f: print(d, "no output"); continue
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    big = 0
    rows = list(csv.DictReader(open(f[0])))
    for r in rows:
        if "k_best8" in r["Kernel_Name"]: big = max(big, int(r["Grid_Size"]))
    for r in rows:
        if "k_best8" in r["Kernel_Name"] and int(r["Grid_Size"]) == big:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in agg: print(d, k, agg[k] / len(n[k]))
PY
tail -3 gpurun_out/ic1.log | cut -c1-300
