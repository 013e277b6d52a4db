#!/bin/bash
# dev: the GPU suite and a fuzz run under every table path (direct atomics / bucketed) and both partition element formats
for p in direct bucket auto; do
  for pk in 1 0; do
    echo "== MODGPU_TABLE_PATH=$p MODGPU_PART_PACKED=$pk"
    MODGPU_TABLE_PATH=$p MODGPU_PART_PACKED=$pk python -m pytest tests -q -m gpu -x 2>&1 | tail -3
    MODGPU_TABLE_PATH=$p MODGPU_PART_PACKED=$pk python tests/fuzz_gpu.py $((7 + pk)) 150 2>&1 | tail -2
  done
done
