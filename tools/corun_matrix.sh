#!/bin/bash
# dev: scan || build co-run (tools/corun_probe.py) under combinations of the build's launch shape and the scan's residency.
# Each combination is a fresh process (the knobs are read once).  usage: tools/corun_matrix.sh > gpurun_out/corun.txt
run () { echo "== $*"; env "$@" python tools/corun_probe.py 2>&1 | tail -1; }
run X=base
run MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512
run MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512 MODGPU_SCAN_GRID=16384
run MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512 MODGPU_SCAN_GRID=8192
run MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512 CORUN_BUILD_PRIORITY=-1
run MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=256
run MODGPU_BUCKET_R=1024 MODGPU_BUCKET_T=256
run MODGPU_BUCKET_R=1024 MODGPU_BUCKET_T=256 MODGPU_SCAN_GRID=16384
if [ -f tools/variants/part512/libmodgpu.so ]; then
  L=$PWD/tools/variants/part512/libmodgpu.so
  run MODGPU_LIB=$L MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512
  run MODGPU_LIB=$L MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512 MODGPU_SCAN_GRID=16384
  run MODGPU_LIB=$L MODGPU_BUCKET_R=1024 MODGPU_BUCKET_T=256
fi
