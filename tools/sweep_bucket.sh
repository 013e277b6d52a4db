#!/bin/bash
# dev: the bench under bucket sizes / dedup workgroup sizes (MODGPU_BUCKET_R, MODGPU_BUCKET_T): step and bucket kernels per setting
for cfg in "1024 512" "1024 1024" "2048 1024" "4096 1024"; do
  set -- $cfg
  MODGPU_BUCKET_R=$1 MODGPU_BUCKET_T=$2 python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py "R=$1,T=$2" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgPartScatterKernel': [0-9.]*" | tr '\n' ' '; echo
done
python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py "default" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgPartScatterKernel': [0-9.]*" | tr '\n' ' '; echo
