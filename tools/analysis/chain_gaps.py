#!/usr/bin/env python3
"""The kernel chain of ONE batch inside the pipelined window of `bench.py`: from a `rocprofv3 --kernel-trace` CSV, the kernels
of one hardware queue between two consecutive main walks, each with its duration and the gap since the previous kernel of the same
queue ended -- where a step's dependent chain spends its time (kernels vs launch gaps).
    python tools/analysis/chain_gaps.py gpurun_out/<dir>/p_kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
by_q = defaultdict(list)
for r in rows:
    by_q[r[qkey] if qkey else "0"].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[-60:]))
for q, ev in sorted(by_q.items(), key=lambda kv: -len(kv[1]))[:2]:
    ev.sort()
    # the main walk: k_best8<STATS = false, LBITS, ARG = false, TIES = false[, B3]> (the coarse pass is the ARG variant)
    walks = [i for i, (s, e, n) in enumerate(ev) if re.search(r"k_best8<false, \w+, false, false(, \w+)?>", n) and e - s > 500_000]
    if len(walks) < 6:
        continue
    # from the end of one main walk to the end of the next: one batch -- a pair inside the pipelined window (the trace also holds the
    # bench's other sections: pairs across their boundaries are milliseconds apart), the one of median length among those
    # (... and not one of the bench's host-buffer sections: no upload-side kernel between the two walks)
    pairs = sorted((ev[walks[k + 1]][1] - ev[walks[k]][1], k) for k in range(len(walks) - 1)
                   if not any("k_rows_prepare" in n or "k_copy_words" in n for _, _, n in ev[walks[k]:walks[k + 1]]))
    near = [p for p in pairs if p[0] < 2 * pairs[0][0]]
    k = near[len(near) // 2][1]
    a, b = walks[k], walks[k + 1]
    print("queue %s: %d kernels, %d main walks; one batch (walk end -> next walk end) = %.3f ms" % (q, len(ev), len(walks), (ev[b][1] - ev[a][1]) / 1e6))
    tk = tg = 0
    prev_end = ev[a][1]
    for s, e, n in ev[a + 1:b + 1]:
        print("   gap %7.1f us   run %8.1f us   %s" % ((s - prev_end) / 1e3, (e - s) / 1e3, n))
        tk += e - s
        tg += max(0, s - prev_end)
        prev_end = e
    print("   kernels %.3f ms, gaps %.3f ms" % (tk / 1e6, tg / 1e6))
