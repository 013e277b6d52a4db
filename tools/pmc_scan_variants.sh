#!/bin/bash
# dev: instruction counts of the scan kernel for library variants (one --pmc pass each): tools/pmc_scan_variants.sh name...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in "$@"; do
  export MODGPU_LIB=$R/tools/variants/$n/libmodgpu.so
  rm -rf /tmp/pmcv_$n; mkdir -p /tmp/pmcv_$n
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmcv_$n/sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-other > /tmp/pmcv_$n/log 2>&1)
  echo "== $n"; python3 $R/tools/pmc_summary.py /tmp/pmcv_$n 2>/dev/null | grep "^mgScanKernel"
done
