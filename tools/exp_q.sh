for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu --no-other 2>/dev/null | grep "^{" | python tools/kern_ms.py "$name" | sed "s/'PartChunks+Scan'.*'PartScatter'/'PartScatter'/; s/'RankScan'.*//; s/'SegScan'.*//"
done
