#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/prof_pmc.sh: per-kernel average duration (kernel trace), per-launch HBM
bytes from FETCH_SIZE / WRITE_SIZE (KiB units) and per-launch SQ / GRBM counters.

FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md §HBM): the
doubling applies to kernels whose reads are such streams; for gather-dominated kernels (random 16-byte rank records,
random bucket probes) the raw figure is the one to use.  Both are printed; STREAMING names which kernels get x2 in
profiles/traffic.json."""
import csv, glob, os, sys, json, collections
root = sys.argv[1]
STREAMING = {"mgScanKernel", "mgSegCompactKernel", "mgPartHistKernel", "mgPartHistBytesKernel", "mgPartScatterKernel", "mgRankCountKernel",
             "mgSynthReadsKernel", "mgSynthGenomeKernel", "mgBucketFindKernel", "mgTableHistKernel", "mgTileInfoKernel", "mgPackKernel", "mgUnpackKernel"}
def find(sub, pat):
    fs = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return fs[0] if fs else None
def kname(s):
    s = s.split("(")[0].replace("void ", "")
    return s.split("<")[0]
out = {}
ks = find("trace", "*kernel_stats.csv")
if ks:
    for r in csv.DictReader(open(ks)):
        name = kname(r["Name"])
        o = out.setdefault(name, {})
        o["total_ms"] = o.get("total_ms", 0.0) + float(r["TotalDurationNs"]) / 1e6 if "TotalDurationNs" in r else o.get("total_ms", 0.0)
        o["calls"] = o.get("calls", 0) + int(r["Calls"])
        o["avg_ms"] = (o["total_ms"] / o["calls"]) if o.get("total_ms") else float(r["AverageNs"]) / 1e6
for sub in ("fetch", "write", "sq", "sq2", "grbm"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        for c, vals in cs.items():
            vals = vals[len(vals) // 2:] if len(vals) > 2 else vals      # skip warm-up launches
            out.setdefault(name, {})[c] = sum(vals) / len(vals)
for name, d in sorted(out.items(), key=lambda kv: -kv[1].get("avg_ms", 0)):
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        fk = d.get("FETCH_SIZE", 0.0); wk = d.get("WRITE_SIZE", 0.0)
        d["hbm_bytes_raw"] = (fk + wk) * 1024
        d["hbm_bytes_fetch_x2"] = (2 * fk + wk) * 1024
        d["hbm_bytes"] = d["hbm_bytes_fetch_x2"] if name in STREAMING else d["hbm_bytes_raw"]
        d["fetch_correction"] = "x2 (wide coalesced streaming reads)" if name in STREAMING else "raw (gather-dominated or mixed)"
    print(name, json.dumps(d))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
