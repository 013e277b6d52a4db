#!/bin/bash
# dev: scan || build co-run (tools/corun_probe.py) under combinations of the build's launch shape, the scan's residency
# (MODGPU_SCAN_GRID workers = wavefronts: 4096 = 4 per SIMD) and the build waves' issue priority.  Each combination is a
# fresh process (the knobs are read once).
run () { echo "== $*"; env "$@" python tools/corun_probe.py 2>&1 | tail -1; }
for g in 4096 3072 2048; do
  run MODGPU_SCAN_GRID=$g
  run MODGPU_SCAN_GRID=$g MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512
  run MODGPU_SCAN_GRID=$g MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=256
  run MODGPU_SCAN_GRID=$g MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512 MODGPU_LIB=$PWD/tools/variants/part512/libmodgpu.so
done
run MODGPU_SCAN_GRID=4096 MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512 MODGPU_LIB=$PWD/tools/variants/noprio/libmodgpu.so
