#!/bin/bash
# rocprofv3 kernel-trace summary and HBM traffic of -p at scale (ugp_scores_per_node, k_scores_level); run on the GPU box
# through gpurun.  Usage: bash tools/profile_scores.sh r03 [samples]
TAG=${1:-r03}
Q=${2:-128}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 tools/bench_scores.py 10000000 $Q"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_scores_stats -o p --output-format csv -- $B > gpurun_out/${TAG}_scores_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_scores_fetch -o p --output-format csv -- $B > gpurun_out/${TAG}_scores_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_scores_write -o p --output-format csv -- $B > gpurun_out/${TAG}_scores_write.log 2>&1
grep -E "scores_per_node|roofline" gpurun_out/${TAG}_scores_stats.log
python3 - <<PY
import csv, glob, collections
tag = "${TAG}"
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob("gpurun_out/%s_scores_%s/*counter_collection.csv" % (tag, sub))
    if not f: continue
    tot = collections.defaultdict(float); nd = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        if "k_scores_level" in r["Kernel_Name"] and r["Counter_Name"] == name:
            tot[name] += float(r["Counter_Value"]); nd[name].add(r["Dispatch_Id"])
    # the timed call is the last one (the tool first runs a 2-sample warm-up): report all dispatches together
    print(name, "summed over", len(nd[name]), "k_scores_level dispatches:", tot[name], "(KB; reads x2 on gfx950, see profiles/README.md)")
rows = list(csv.DictReader(open(glob.glob("gpurun_out/%s_scores_stats/*kernel_trace.csv" % tag)[0])))
lv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])) for r in rows if "k_scores_level" in r["Kernel_Name"]]
lv.sort()
half = len(lv) // 2   # (second half = the timed call when the warm-up had as many levels)
sel = lv[-(len(lv) - half):] if half else lv
busy = sum(e - s for s, e, g in sel); span = sel[-1][1] - sel[0][0]
print("timed call: %d level launches, kernels busy %.3f ms of a %.3f ms span; the 10 largest launches: %.3f ms" % (len(sel), busy / 1e6, span / 1e6, sum(sorted((e - s for s, e, g in sel), reverse=True)[:10]) / 1e6))
PY
