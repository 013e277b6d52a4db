#!/bin/bash
# dev: A/B of environment settings on the headline workload, one bench run per setting, step and kernels on one line each (round 6's table-geometry
# sweeps: profiles/r06_ab_table_geometry.txt).  usage: tools/ab_env.sh name:VAR=value[ VAR2=value2] ...   e.g.
#   tools/ab_env.sh base: t60:MODGPU_TIGHT_LOAD=60 "s1:MODGPU_MERGE_SLOTS=1 MODGPU_FLAG_POLARITY=1"
# With ABL=1 the -DMG_ABLATE build under tools/variants_abl/ is used (tools/ablate_build.sh), so that MODGPU_BUCKET_DEBUG bits take effect.
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  [ -n "$ABL" ] && envs="$envs MODGPU_LIB=$PWD/tools/variants_abl/libmodgpu.so"
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu --no-other 2>/dev/null | grep "^{" | python tools/kern_ms.py "$name" | sed "s/'PartChunks+Scan'[^']*//; s/'RankScan'.*//; s/'SegScan'.*//"
done
