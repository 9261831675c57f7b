#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_*.
   python tools/summarize_profile.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
stats = glob.glob(os.path.join(go, tag + "_stats", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(prof, tag + "_kernel_stats.csv"))
summary = {"tag": tag, "command": "python3 bench.py --queries 16384 --steps 3 --warmup 1 --cpu-queries 0 --no-extra --repeats 1" + (" " + sys.argv[2] if len(sys.argv) > 2 else ""), "kernels": {}}
log = os.path.join(go, tag + "_stats.log")
if os.path.exists(log):
    with open(log) as f:
        lines = [l for l in f if l.startswith("{")]
    if lines:
        summary["bench"] = json.loads(lines[-1])
for sub in ("fetch", "write", "tcc", "sq", "sq2"):
    files = glob.glob(os.path.join(go, "%s_%s" % (tag, sub), "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    rows = list(csv.DictReader(open(files[0])))
    # a step launches some kernels twice (coarse locality pass + full pass): keep the full-size dispatches
    biggest = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        biggest[k] = max(biggest[k], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Grid_Size"]) != biggest[k]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in agg:
        if not k.startswith("ugp::") and "ugp::" not in k:
            continue
        e = summary["kernels"].setdefault(k, {})
        for c, v in agg[k].items():
            e[c + "_per_dispatch"] = v / len(disp[k])
# average duration of the full-size dispatches from the kernel trace (the stats CSV mixes both sizes)
trace = glob.glob(os.path.join(go, tag + "_stats", "*kernel_trace.csv"))
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    biggest = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        biggest[k] = max(biggest[k], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))
    dur = collections.defaultdict(list)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == biggest[k]:
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in dur.items():
        if "ugp::" in k:
            summary["kernels"].setdefault(k, {})["avg_duration_ns_full_dispatch"] = sum(v) / len(v)
            summary["kernels"][k]["full_dispatches"] = len(v)
for k, e in summary["kernels"].items():
    # gfx950: FETCH_SIZE is in KB and reports half of the bytes of a coalesced streaming read
    # (MI355X_MICROARCH.md, HBM section): double it; WRITE_SIZE is taken as reported (KB).
    if "FETCH_SIZE_per_dispatch" in e:
        e["hbm_read_bytes_per_dispatch_corrected"] = e["FETCH_SIZE_per_dispatch"] * 1024 * 2
    if "WRITE_SIZE_per_dispatch" in e:
        e["hbm_write_bytes_per_dispatch"] = e["WRITE_SIZE_per_dispatch"] * 1024
    if "TCC_HIT_sum_per_dispatch" in e:
        e["l2_hit_rate"] = e["TCC_HIT_sum_per_dispatch"] / (e["TCC_HIT_sum_per_dispatch"] + e["TCC_MISS_sum_per_dispatch"])
    # Instruction issue (the bound that applies to k_best8).  A wave64 VALU instruction occupies its SIMD-32 for 2
    # cycles (MI355X_MICROARCH.md "Wave scheduling"): chip capacity = 256 CU x 4 SIMD x f x t / 2; the scalar unit
    # is one per CU: 256 x f x t issue slots.  f = 2.4 GHz nominal.
    if "avg_duration_ns_full_dispatch" in e:
        t = e["avg_duration_ns_full_dispatch"] * 1e-9
        if "SQ_INSTS_VALU_per_dispatch" in e:
            e["valu_issue_frac"] = e["SQ_INSTS_VALU_per_dispatch"] / (256 * 4 * 2.4e9 * t / 2)
        if "SQ_INSTS_SALU_per_dispatch" in e:
            e["salu_issue_frac"] = (e["SQ_INSTS_SALU_per_dispatch"] + e.get("SQ_INSTS_SMEM_per_dispatch", 0.0)) / (256 * 2.4e9 * t)
    if "SQ_INST_CYCLES_SALU_per_dispatch" in e and e.get("SQ_BUSY_CU_CYCLES_per_dispatch"):
        e["salu_busy_frac_measured"] = e["SQ_INST_CYCLES_SALU_per_dispatch"] / e["SQ_BUSY_CU_CYCLES_per_dispatch"]
    # SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave has a vector instruction in flight: x4 / (SIMD-cycles of the
    # dispatch) = how busy the vector pipes were by the SQ's own account (tools/micro/issue_rate.hip: plain 32-bit VALU
    # issues every 2 cycles per SIMD, v_pk_*_u16 and v_readlane every 4)
    if "SQ_ACTIVE_INST_VALU_per_dispatch" in e and "avg_duration_ns_full_dispatch" in e:
        e["valu_active_frac_measured"] = e["SQ_ACTIVE_INST_VALU_per_dispatch"] * 4 / (256 * 4 * 2.4e9 * e["avg_duration_ns_full_dispatch"] * 1e-9)
    if "SQ_WAIT_ANY_per_dispatch" in e and e.get("SQ_WAVE_CYCLES_per_dispatch"):
        e["wave_wait_frac"] = e["SQ_WAIT_ANY_per_dispatch"] / e["SQ_WAVE_CYCLES_per_dispatch"]
with open(os.path.join(prof, tag + "_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)
print(json.dumps({k: {c: round(v, 3) for c, v in e.items()} for k, e in summary["kernels"].items() if "best8" in k}, indent=1))
