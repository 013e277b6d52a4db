mkdir -p gpurun_out/r6c
true
for cfg in "t0:MODGPU_TIGHT_LOAD=0" "t50:MODGPU_TIGHT_LOAD=50" "t60:MODGPU_TIGHT_LOAD=60" "t70:MODGPU_TIGHT_LOAD=70" "t65:MODGPU_TIGHT_LOAD=65" "t80:MODGPU_TIGHT_LOAD=80"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu --no-other > gpurun_out/r6c/$name.out 2> gpurun_out/r6c/$name.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r6c/$name.out") if x.startswith("{")]
if not l: print("$name", "FAILED", open("gpurun_out/r6c/$name.err").read()[-600:])
else:
  j=json.loads(l[-1]); r=j["roofline"]
  print("$name", j["value"], j["ms_per_step"], json.dumps(r["kernels_ms_per_step"]))
PY
done
