for tl in "$@"; do
  MODGPU_TABLE_LOAD=$tl python bench.py --only c3 --steps 5 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())['result']
print('load', $tl, j['value'], j['ms_per_batch'], {k[2:-6]: round(v, 3) for k, v in j['roofline']['kernels_ms_per_step'].items() if v > 0.3})"
done
