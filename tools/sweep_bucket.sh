#!/bin/bash
for cfg in "2048 256" "2048 512" "4096 512" "4096 1024" "8192 1024" "8192 512"; do
  set -- $cfg
  MODGPU_BUCKET_R=$1 MODGPU_BUCKET_T=$2 python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python tools/kern_ms.py "R=$1,T=$2"
done
