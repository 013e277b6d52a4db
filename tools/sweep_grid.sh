#!/bin/bash
for g in 6144 9216 12288 18432 24576 49152; do
  MODGPU_SCAN_GRID=$g python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py grid=$g | grep -o "^.*ms/step\|'mgScanKernel': [0-9.]*\|'mgSegCompactKernel': [0-9.]*" | tr '\n' ' '; echo
done
