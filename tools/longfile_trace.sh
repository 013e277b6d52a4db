#!/bin/bash
# dev: kernel stats of bench.py's file legs (end_to_end.modmap_query_file_long and the others), to see which of the text parser's kernels the
# reference / query file reads spend their time in -> gpurun_out/longfile_kernel_stats.csv
export TMPDIR=/tmp MODGPU_BENCH_OTHER=none MODGPU_CPU_SAMPLE_MBP=50 MODGPU_E2E_GBP=0.3
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/lf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lf -- python3 $R/bench.py --steps 2 --warmup 1 > /tmp/lf.log 2>&1
cd $R
cp $(ls -S $(find /tmp/lf -name "*kernel_stats.csv") | head -1) gpurun_out/longfile_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/longfile_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:22]:
    print("%-62s calls %6s avg %9.3f ms total %9.2f ms" % (r['Name'][:62], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
