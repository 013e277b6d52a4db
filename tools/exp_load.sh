mkdir -p gpurun_out/r6b
for cfg in "base:" "load120:MODGPU_TABLE_LOAD=120" "load120s0:MODGPU_TABLE_LOAD=120 MODGPU_MERGE_SLOTS=0" "load120s1:MODGPU_TABLE_LOAD=120 MODGPU_MERGE_SLOTS=1" "load100:MODGPU_TABLE_LOAD=100" "s1:MODGPU_MERGE_SLOTS=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu --no-other > gpurun_out/r6b/$name.out 2> gpurun_out/r6b/$name.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r6b/$name.out") if x.startswith("{")]
j=json.loads(l[-1]); r=j["roofline"]
print("$name", j["value"], j["ms_per_step"], json.dumps(r["kernels_ms_per_step"]))
PY
done
