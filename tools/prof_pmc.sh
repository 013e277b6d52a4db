#!/bin/bash
# tools/prof_pmc.sh <tag> [bench args] — kernel-trace stats + separate PMC passes for bench.py
# (counters are collected in their own runs, --pmc with --kernel-trace only, as MI355X_MICROARCH.md prescribes;
#  the program itself follows `--`: no env / bash -c hop under the profiler)
TAG=${1:-r02}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu --no-other}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
# instruction issue of the kernels (8 SQ slots per pass), and the busy-cycle / clock counters on their own
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -- python3 $R/bench.py $ARGS > $OUT/grbm.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
