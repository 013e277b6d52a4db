#!/bin/bash
# dev: what bounds mgRankLookupKernel?  MG_ABLATE build, MODGPU_BUCKET_DEBUG: 16 no rank gathers, 128 no index stores, 256 no
# ordinal loads (results are wrong in those runs; only the kernel time is of interest)
for dbg in 0 16 128 256 144 272 400; do
  echo -n "debug $dbg: "
  MODGPU_LIB=$PWD/tools/variants/abl/libmodgpu.so MODGPU_BUCKET_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --no-cpu --no-other 2>/dev/null | python tools/kern_ms.py x | grep -o "'mgRankLookupKernel': [0-9.]*\|'mgBucketMergeKernel': [0-9.]*"  | tr '\n' ' '; echo
done
