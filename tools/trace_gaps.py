#!/usr/bin/env python3
"""dev: idle time between the kernels of one batch, from a rocprofv3 --kernel-trace CSV: tools/trace_gaps.py <dir> <first kernel's name part>
prints the kernels from one launch of that kernel to the next (the last complete such span) with the gap in front of each."""
import csv, glob, sys
d, first = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; busy = 0; gaps = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    print("%9.1f us  gap %7.1f  run %8.1f  %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    busy += e - s; gaps += max(0, gap); prev_end = max(prev_end, e)
print("span %.1f us: kernels %.1f, gaps %.1f; to the next first kernel %.1f" % ((prev_end - t0) / 1e3, busy / 1e3, gaps / 1e3, (int(rows[b]["Start_Timestamp"]) - prev_end) / 1e3))
