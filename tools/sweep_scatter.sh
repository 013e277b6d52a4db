#!/bin/bash
# dev: the bench under grids of the partition scatter kernel (MODGPU_SCATTER_GRID): step and scatter time per setting
for g in 256 512 1024 2048 100000; do
  MODGPU_SCATTER_GRID=$g python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py sg=$g | grep -o "^.*ms/step\|'mgPartScatterKernel': [0-9.]*" | tr '\n' ' '; echo
done
