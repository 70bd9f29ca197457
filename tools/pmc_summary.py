"""Summarise rocprofv3 --pmc output (…_counter_collection.csv files under a directory): per kernel and counter, launches,
mean value and mean duration.  usage: python tools/pmc_summary.py DIR [name-filter] > profiles/xyz.json  (measurement tool)"""
import csv, glob, json, os, re, sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        if flt and flt not in name:
            continue
        k = acc.setdefault(name, {"vgpr": int(r["VGPR_Count"]), "scratch": int(r["Scratch_Size"]), "lds": int(r["LDS_Block_Size"]),
                                  "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "counters": {}})
        c = k["counters"].setdefault(r["Counter_Name"], {"n": 0, "sum": 0.0, "ns": 0.0})
        c["n"] += 1
        c["sum"] += float(r["Counter_Value"])
        c["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for name, k in sorted(acc.items()):
    out[name] = {kk: vv for kk, vv in k.items() if kk != "counters"}
    for cn, c in sorted(k["counters"].items()):
        out[name][cn] = {"launches": c["n"], "mean": c["sum"] / c["n"], "mean_ms_under_profiler": c["ns"] / c["n"] / 1e6}
print(json.dumps(out, indent=1))
