#!/bin/bash
# dev: config 3 alone, kernel table
for i in ${@:-1 2}; do MODGPU_BENCH_OTHER=c3 python bench.py --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['other_configs']['c3']; print('c3', c['value'], c['ms_per_batch'], c['roofline']['kernels_ms_per_step'])"; done
