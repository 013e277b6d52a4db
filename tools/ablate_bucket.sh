#!/bin/bash
# dev: ablation timing of the bucket / lookup kernels on a -DMG_ABLATE build (built here, on the CPU box, into
# tools/variants_abl/ so that it travels: L=$(bash tools/ablate_build.sh) && mkdir -p tools/variants_abl && cp $L tools/variants_abl/).
# bits of MODGPU_BUCKET_DEBUG: see MgBucketArgs.debug in mg_table.hip
for dbg in ${@:-0 1 2 4 8 16 128 256 400 32 64}; do
  MODGPU_BUCKET_DEBUG=$dbg MODGPU_LIB=$PWD/tools/variants_abl/libmodgpu.so python bench.py --steps 5 --warmup 1 --no-cpu --no-other 2>/dev/null | python tools/kern_ms.py "dbg=$dbg" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgRank[A-Za-z]*': [0-9.]*" | tr '\n' ' '; echo
done
