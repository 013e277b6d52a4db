for tl in 0 60; do
for dbg in 0 32 64 96 4 8 1 13; do
  MODGPU_TIGHT_LOAD=$tl MODGPU_BUCKET_DEBUG=$dbg MODGPU_LIB=$PWD/tools/variants_abl/libmodgpu.so python bench.py --steps 5 --warmup 1 --no-cpu --no-other 2>/dev/null | grep "^{" | python tools/kern_ms.py "tight=$tl dbg=$dbg" | grep -o "^.*ms/step\|'BucketDedup': [0-9.]*\|'BucketMerge': [0-9.]*" | tr '\n' ' '; echo
done; done
