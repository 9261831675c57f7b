#!/usr/bin/env python3
"""What the device does while consecutive ugp_place_device calls overlap: from a `rocprofv3 --kernel-trace` CSV of
`bench.py --steps 20`, the share of the timed window with 0 / 1 / 2+ kernels running, with 0 / 1 / 2 tree walks (k_best8)
running, and the summed duration per kernel per step.
    python tools/analysis/pipeline_timeline.py gpurun_out/<dir>/p_kernel_trace.csv [steps]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
walks = [(s, e) for s, e, n in ev if re.search(r"k_best8<false, \w+, false, false(, \w+)?>", n) and e - s > 600_000]   # main walks (not the coarse pass: ARG)
# the timed loop: the longest run of main walks that start less than 3 ms apart
best = (0, 0)
i = 0
while i < len(walks):
    j = i
    while j + 1 < len(walks) and walks[j + 1][0] - walks[j][0] < 3_000_000:
        j += 1
    if j - i > best[1] - best[0]:
        best = (i, j)
    i = j + 1
lo, hi = walks[best[0]][0], walks[best[1]][1]
n_walks = best[1] - best[0] + 1
print("window: %d main walks back to back, %.2f ms = %.3f ms per walk" % (n_walks, (hi - lo) / 1e6, (hi - lo) / 1e6 / n_walks))


def coverage(intervals):
    pts = []
    for s, e in intervals:
        s, e = max(s, lo), min(e, hi)
        if s < e:
            pts += [(s, 1), (e, -1)]
    pts.sort()
    hist = defaultdict(int)
    depth, last = 0, lo
    for t, d in pts:
        hist[depth] += t - last
        last = t
        depth += d
    hist[depth] += hi - last
    return hist


tot = hi - lo
h = coverage([(s, e) for s, e, n in ev])
print("kernels running:  " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(h.items()) if v))
h = coverage([(s, e) for s, e, n in ev if "k_best8" in n and e - s > 600_000])
print("main walks running: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(h.items()) if v))
per = defaultdict(lambda: [0, 0])
for s, e, n in ev:
    if s >= lo and e <= hi:
        k = n.split("(")[0][-44:]
        per[k][0] += e - s
        per[k][1] += 1
print("summed kernel time per walk (overlapped durations), top 12:")
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %-46s %8.3f ms  (%.1f launches)" % (k, t / 1e6 / n_walks, c / n_walks))
print("  total %.3f ms of kernel time per walk in %.3f ms of wall time" % (sum(t for t, _ in per.values()) / 1e6 / n_walks, tot / 1e6 / n_walks))
