for l in 60 75 90 100; do MODGPU_TABLE_LOAD=$l MODGPU_BENCH_FORCE_DIST=1 python bench.py --no-cpu --no-other --steps 5 --warmup 1 2>/dev/null | python tools/kern_ms.py "c4block load=$l" | grep -o "^.*ms/step\|'mgBucket[A-Za-z]*': [0-9.]*\|'mgRankLookupKernel': [0-9.]*" | tr '\n' ' '; echo; done
for l in 60 75; do MODGPU_TABLE_LOAD=$l MODGPU_BENCH_OTHER=c3 python bench.py --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['other_configs']['c3']; print('c3 load=$l', c['value'], c['roofline']['kernels_ms_per_step'])"; done
