#!/bin/bash
# dev: time prebuilt library variants (tools/variants/<name>/libmodgpu.so, chosen with MODGPU_LIB) with the bench,
# interleaved on one box (boxes differ by a few per cent).  usage: ab_variants.sh [names...] (default: all)
names=${@:-$(ls tools/variants)}
for rep in 1 2 3; do
for n in $names; do
  MODGPU_LIB=$PWD/tools/variants/$n/libmodgpu.so python bench.py --steps 8 --warmup 2 --no-cpu --no-other 2>/dev/null | python tools/kern_ms.py "$n" | sed 's/mgBucket//g; s/Kernel//g; s/mgPart//g; s/mgRank/R/g; s/mg//g'
done
done
