#!/bin/bash
# dev: kernel-trace stats of the bench with an environment setting on and off (per-instantiation kernel times).  usage: trace_ab.sh VAR
VAR=${1:-MODGPU_PART_ALIGNED}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  OUT=$R/gpurun_out/trace_ab_$v; rm -rf $OUT; mkdir -p $OUT
  export $VAR=$v
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-other > $OUT/trace.log 2>&1
  cd $R
  echo "== $VAR=$v"
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "Scatter" in n or "Hist" in n or "Dedup" in n or "PartScan" in n:
        print("  %-60s calls %3s avg %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/trace
done
