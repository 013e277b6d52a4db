#!/bin/bash
# dev: the dedup kernel without its flag stores, on a -DMG_ABLATE build
export MODGPU_LIB=$(bash "$(dirname "$0")/ablate_build.sh")
for d in 0 1; do
  MODGPU_BUCKET_DEBUG=$d python bench.py --steps 3 --warmup 1 --no-cpu --no-other 2>/dev/null | python tools/kern_ms.py "debug=$d" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgRank[A-Za-z]*': [0-9.]*" | tr '\n' ' '; echo
done
