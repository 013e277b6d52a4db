#!/bin/bash
# dev: where the scan's time goes.  MG_ABLATE build, MODGPU_SCAN_DEBUG: 1 = phases A + B only (filter + candidate count, no
# listing, no evaluation), 4 = no evaluation (candidates listed, nothing computed for them), 2 = no stores
for dbg in 0 1 4 2; do
  echo -n "scan debug $dbg: "
  MODGPU_LIB=$PWD/tools/variants/abl/libmodgpu.so MODGPU_SCAN_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --no-cpu --no-other 2>/dev/null | python tools/kern_ms.py x | grep -o "'mgScanKernel': [0-9.]*"
done
