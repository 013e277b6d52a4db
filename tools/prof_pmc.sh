#!/bin/bash
# tools/prof_pmc.sh <tag> — kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE) for bench.py
# (counters are collected in their own runs, as MI355X_MICROARCH.md's HBM section prescribes)
TAG=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/write.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
