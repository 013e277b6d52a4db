#!/bin/bash
for g in 256 512 768 1024 1536 2048; do
  MODGPU_SCAN_GRID=$g python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py grid=$g
done
