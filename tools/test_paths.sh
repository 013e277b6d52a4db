#!/bin/bash
for p in direct bucket auto; do
  echo "== MODGPU_TABLE_PATH=$p"
  MODGPU_TABLE_PATH=$p python -m pytest tests -q -m gpu -x 2>&1 | tail -4
done
