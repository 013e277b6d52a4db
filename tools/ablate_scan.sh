#!/bin/bash
# dev: phase ablation of mgScanKernel on a -DMG_ABLATE build (bit0 = stop after phase B, bit1 = no stores, bit2 = no evaluation)
export MODGPU_LIB=$(bash "$(dirname "$0")/ablate_build.sh")
for d in 0 1 2 4 6; do
  MODGPU_SCAN_DEBUG=$d python bench.py --steps 3 --warmup 1 --no-cpu --no-other 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('debug=$d scan ms', j['roofline']['kernels_ms_per_step'].get('mgScanKernel'))"
done
