#!/bin/bash
# tools/prof_round.sh <round tag, e.g. r03> — the round's profiles: kernel-trace stats + separate --pmc passes (as
# MI355X_MICROARCH.md prescribes: counters in their own runs, --pmc with nothing else) of bench.py for the headline workload
# (config 2) and, with --only, for configs 3, 5 and one block of config 4.  Results: gpurun_out/prof_<tag>_{c2,c3,c5,c4}/summary.txt
TAG=${1:-r03}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof () {   # name, bench args...
  local name=$1; shift
  local OUT=$R/gpurun_out/prof_${TAG}_$name
  rm -rf $OUT; mkdir -p $OUT
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py "$@" > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py "$@" > $OUT/write.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 $R/bench.py "$@" > $OUT/sq.log 2>&1
  if [ "$name" = c2 ]; then
    rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $R/bench.py "$@" > $OUT/sq2.log 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -- python3 $R/bench.py "$@" > $OUT/grbm.log 2>&1
  fi
  cd $R
  python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
  rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/sq $OUT/sq2 $OUT/grbm       # the raw per-dispatch CSVs are large
  head -c 3000 $OUT/summary.txt
}
ONLY=${2:-all}                                  # second argument: one of c2 c3 c5 c4 (default: all four)
[ $ONLY = all -o $ONLY = c2 ] && prof c2 --steps 5 --warmup 1 --no-cpu --no-other
[ $ONLY = all -o $ONLY = c3 ] && prof c3 --only c3 --steps 3
[ $ONLY = all -o $ONLY = c5 ] && prof c5 --only c5 --steps 5
[ $ONLY = all -o $ONLY = c4 ] && prof c4 --only c4_block --steps 3
true
