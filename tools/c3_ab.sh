# dev: config 3 with the find kernel's grid capped (MODGPU_FIND_GRID)
for g in ${@:-0 512 1024 1536 2048 4096}; do MODGPU_FIND_GRID=$g MODGPU_BENCH_OTHER=c3 python bench.py --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['other_configs']['c3']; print('grid=$g', c['value'], c['roofline']['kernels_ms_per_step'])"; done
