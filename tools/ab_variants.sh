#!/bin/bash
# dev: time prebuilt library variants (tools/variants/*.so) with the bench, interleaved
cp modimizer_amd/libmodgpu.so /tmp/libmodgpu.keep
for rep in 1 2 3; do
for f in tools/variants/*.so; do
  cp $f modimizer_amd/libmodgpu.so
  python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py "$(basename $f)" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgPartScatterKernel': [0-9.]*" | tr '\n' ' '; echo
done
done
cp /tmp/libmodgpu.keep modimizer_amd/libmodgpu.so
