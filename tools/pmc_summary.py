#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/prof_pmc.sh: per-kernel average duration (kernel trace) and
per-launch HBM bytes from FETCH_SIZE / WRITE_SIZE (KiB units; FETCH_SIZE doubled on gfx950 as
MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads — flagged, not silently applied)."""
import csv, glob, os, sys, json, collections
root = sys.argv[1]
def find(sub, pat):
    fs = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return fs[0] if fs else None
out = {}
ks = find("trace", "*kernel_stats.csv")
if ks:
    for r in csv.DictReader(open(ks)):
        name = r["Name"].split("(")[0].replace("void ", "")
        out.setdefault(name, {})["avg_ms"] = float(r["AverageNs"]) / 1e6
        out[name]["calls"] = int(r["Calls"])
for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == key:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name].append(float(r["Counter_Value"]))
    for name, vals in acc.items():
        vals = vals[len(vals) // 2:] if len(vals) > 2 else vals      # skip warm-up launches
        out.setdefault(name, {})[key + "_KiB_per_launch"] = sum(vals) / len(vals)
for name, d in sorted(out.items(), key=lambda kv: -kv[1].get("avg_ms", 0)):
    if "FETCH_SIZE_KiB_per_launch" in d or "WRITE_SIZE_KiB_per_launch" in d:
        fk = d.get("FETCH_SIZE_KiB_per_launch", 0.0); wk = d.get("WRITE_SIZE_KiB_per_launch", 0.0)
        d["hbm_bytes_raw"] = (fk + wk) * 1024
        d["hbm_bytes_fetch_x2"] = (2 * fk + wk) * 1024
    print(name, json.dumps(d))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
