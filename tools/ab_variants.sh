#!/bin/bash
# dev: time prebuilt library variants (tools/variants/<name>/libmodgpu.so, chosen with MODGPU_LIB) with the bench,
# interleaved on one box (boxes differ by a few per cent).  usage: ab_variants.sh [bench env assignments...]
for rep in 1 2 3; do
for d in tools/variants/*/; do
  env "$@" MODGPU_LIB=$PWD/${d}libmodgpu.so python bench.py --steps 5 --warmup 1 --no-cpu --no-other 2>/dev/null | python tools/kern_ms.py "$(basename $d)" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgRankLookupKernel': [0-9.]*\|'mgPartScatterKernel': [0-9.]*\|'mgScanKernel': [0-9.]*" | tr '\n' ' '; echo
done
done
