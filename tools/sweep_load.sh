#!/bin/bash
# dev: the bench under table loads (MODGPU_TABLE_LOAD, per cent of slots per occurrence): step and bucket kernels per setting
for l in 60 150; do
  MODGPU_TABLE_LOAD=$l python bench.py --steps 4 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py load=$l | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*" | tr '\n' ' '; echo
done
