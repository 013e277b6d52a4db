#!/usr/bin/env python3
"""dev: phases of tools/corun_probe.py from a rocprofv3 kernel trace.  The probe runs warm-up, then 6 scans alone, 6 builds
alone, then 6 + 6 at once: a kernel launch belongs to the co-run phase when a kernel of the OTHER family (scan / build) overlaps
it in time.  Prints, per kernel name, launches and mean duration alone and overlapped, and the overlapped share of its time."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda x: x[1])
SCAN = ("mgScanKernel", "mgTileInfoKernel", "mgSegScanKernel", "mgSegCompactKernel")
fam = lambda n: "scan" if n in SCAN else ("build" if n.startswith("mg") else "other")
ivals = {"scan": [], "build": []}
for n, a, b in rows:
    if fam(n) in ivals: ivals[fam(n)].append((a, b))
def overlap(a, b, other):
    tot = 0
    for x, y in other:
        if y <= a: continue
        if x >= b: break
        tot += min(b, y) - max(a, x)
    return tot
stat = collections.defaultdict(lambda: [0, 0.0, 0, 0.0, 0.0])
for n, a, b in rows:
    f = fam(n)
    if f == "other": continue
    ov = overlap(a, b, ivals["build" if f == "scan" else "scan"])
    s = stat[n]
    if ov > 0.5 * (b - a): s[2] += 1; s[3] += (b - a) / 1e6; s[4] += ov / 1e6
    elif ov == 0: s[0] += 1; s[1] += (b - a) / 1e6
print("%-28s %8s %10s %8s %10s %8s" % ("kernel", "n alone", "ms alone", "n co-run", "ms co-run", "slowdown"))
for n, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    if s[0] and s[2]:
        print("%-28s %8d %10.3f %8d %10.3f %8.2f" % (n, s[0], s[1] / s[0], s[2], s[3] / s[2], (s[3] / s[2]) / (s[1] / s[0])))
