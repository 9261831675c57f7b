"""Summary of tools/pmc_fitch.sh's counter passes:  python3 tools/analysis/fitch_pmc.py <tag>   (reads gpurun_out/<tag>_pmc_*)"""
import csv, glob, collections, sys
tag = sys.argv[1]
val = collections.defaultdict(dict)   # (kernel, dispatch) -> counter -> value
dur = {}
for d in glob.glob("gpurun_out/%s_pmc_*/" % tag):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        nm = r["Kernel_Name"]
        if "k_fs_" not in nm: continue
        k = nm.split("k_fs_")[1].split("(")[0]
        key = (k, int(r["Grid_Size"]))
        val[key][r["Counter_Name"]] = val[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        val[key]["_n_" + r["Counter_Name"]] = val[key].get("_n_" + r["Counter_Name"], 0) + 1
    if "SQ_WAVES" in d:
        seen = set()
        for r in csv.DictReader(open(f[0])):
            nm = r["Kernel_Name"]
            if "k_fs_" not in nm or r["Dispatch_Id"] in seen: continue
            seen.add(r["Dispatch_Id"])
            k = nm.split("k_fs_")[1].split("(")[0]
            key = (k, int(r["Grid_Size"]))
            dur.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
def per(key, c):
    v = val[key]
    return v.get(c, 0.0) / max(v.get("_n_" + c, 1), 1)
tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
rows = []
for key in val:
    rd, wr = per(key, "FETCH_SIZE") * 1024 * 2, per(key, "WRITE_SIZE") * 1024
    t = sum(dur.get(key, [0])) / max(len(dur.get(key, [1])), 1)
    n_launch = len(dur.get(key, [1]))
    tot[key[0]][0] += rd * n_launch; tot[key[0]][1] += wr * n_launch; tot[key[0]][2] += t * n_launch
    rows.append((key, rd, wr, t))
for k in ("forward", "backward", "init", "scatter", "mark"):
    big = sorted([r for r in rows if r[0][0] == k], key=lambda r: -r[0][1])[:3]
    for key, rd, wr, t in big:
        wc, wt = per(key, "SQ_WAVE_CYCLES"), per(key, "SQ_WAIT_INST_ANY")
        print("%-9s grid %9d  read %7.1f MB  written %7.1f MB  %7.1f us  %5.2f TB/s  waves %8d  wait %.2f  VALU insts %.3g  SALU %.3g  SMEM %.3g" % (
            k, key[1], rd / 1e6, wr / 1e6, t / 1e3, (rd + wr) / max(t, 1) / 1e3, per(key, "SQ_WAVES"), wt / max(wc, 1), per(key, "SQ_INSTS_VALU"),
            per(key, "SQ_INSTS_SALU"), per(key, "SQ_INSTS_SMEM")))
for k, (rd, wr, t) in tot.items():
    print("all %-9s read %8.1f MB  written %8.1f MB  %8.1f us  %5.2f TB/s" % (k, rd / 1e6, wr / 1e6, t / 1e3, (rd + wr) / max(t, 1) / 1e3))
