#!/bin/bash
# dev: the bench under numbers of scan workers (MODGPU_SCAN_GRID): step, scan and compaction per setting
for g in 16384 32768 49152 98304 196608; do
  MODGPU_SCAN_GRID=$g python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py workers=$g | grep -o "^.*ms/step\|'mgScanKernel': [0-9.]*\|'mgSegCompactKernel': [0-9.]*" | tr '\n' ' '; echo
done
