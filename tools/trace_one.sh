#!/bin/bash
# dev: kernel-trace stats of the bench under the current environment (per-instantiation kernel times).  usage: trace_one.sh <label> [filter regex]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_one_$$; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-other > $OUT/trace.log 2>&1
cd $R
echo "== $1"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${2:-Scatter|Hist|Dedup}" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if re.search(sys.argv[2], n):
        print("  %-60s calls %3s avg %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT
