#!/bin/bash
# dev: the GPU suite and a fuzz run under every table path (direct atomics / bucketed) and both partition element formats
for p in direct bucket auto; do
  for pk in 1 0; do
    echo "== MODGPU_TABLE_PATH=$p MODGPU_PART_PACKED=$pk"
    MODGPU_TABLE_PATH=$p MODGPU_PART_PACKED=$pk python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py 2>&1 | tail -3
    MODGPU_TABLE_PATH=$p MODGPU_PART_PACKED=$pk python tests/fuzz_gpu.py $((7 + pk)) 150 2>&1 | tail -2
  done
done
# the scan's worker geometry: few workers (many tiles per worker: the candidate queue carries across tiles all the time), and the exact-mode kernel for everything
for g in 3 50; do
  echo "== MODGPU_SCAN_GRID=$g"
  MODGPU_SCAN_GRID=$g python -m pytest tests/test_gpu_scan.py tests/test_gpu_modset.py -q -x 2>&1 | tail -2
  MODGPU_SCAN_GRID=$g python tests/fuzz_gpu.py 11 150 2>&1 | tail -2
done
echo "== MODGPU_SCAN_GENERIC=1"
MODGPU_SCAN_GENERIC=1 python -m pytest tests/test_gpu_scan.py tests/test_gpu_modset.py -q -x 2>&1 | tail -2
echo "== MODGPU_PART_BIG=0"
MODGPU_PART_BIG=0 MODGPU_TABLE_PATH=bucket python -m pytest tests/test_gpu_modset.py -q -x 2>&1 | tail -2
MODGPU_PART_BIG=0 MODGPU_TABLE_PATH=bucket python tests/fuzz_gpu.py 13 150 2>&1 | tail -2
echo "== MODGPU_HOT_SPLIT=200,64 (every bucket beyond 200 occurrences through the chunk-wise reduction)"
for pk in 1 0; do
  MODGPU_HOT_SPLIT=200,64 MODGPU_TABLE_PATH=bucket MODGPU_PART_PACKED=$pk python -m pytest tests/test_gpu_modset.py -q -x 2>&1 | tail -2
  MODGPU_HOT_SPLIT=200,64 MODGPU_TABLE_PATH=bucket MODGPU_PART_PACKED=$pk python tests/fuzz_gpu.py $((17 + pk)) 150 2>&1 | tail -2
done
echo "== MODGPU_FIND_PATH=part (every lookup batch through the partitioned path)"
MODGPU_FIND_PATH=part python -m pytest tests/test_gpu_modset.py tests/test_dropin.py tests/test_ref_files.py tests/test_readset.py tests/test_seqio.py -q -x -m gpu 2>&1 | tail -2
MODGPU_FIND_PATH=part python tests/fuzz_gpu.py 19 150 2>&1 | tail -2
echo "== MODGPU_FIND_PATH=2 (two-level partitioned lookups)"
MODGPU_FIND_PATH=2 python -m pytest tests/test_gpu_modset.py tests/test_dropin.py tests/test_ref_files.py tests/test_readset.py tests/test_seqio.py -q -x -m gpu 2>&1 | tail -2
MODGPU_FIND_PATH=2 python tests/fuzz_gpu.py 20 150 2>&1 | tail -2
echo "== MODGPU_RANK_SLICE_SHIFT=16 (the smallest rank-lookup slices, as many as the list groups allow)"
MODGPU_RANK_SLICE_SHIFT=16 MODGPU_TABLE_PATH=bucket python -m pytest tests/test_gpu_modset.py -q -x 2>&1 | tail -2
MODGPU_RANK_SLICE_SHIFT=16 MODGPU_TABLE_PATH=bucket python tests/fuzz_gpu.py 23 150 2>&1 | tail -2
echo "== MODGPU_PART_DIGITS=0 (the second partition pass counts its digits from the elements, not from the first pass's digit bytes)"
MODGPU_PART_DIGITS=0 MODGPU_TABLE_PATH=bucket python -m pytest tests/test_gpu_modset.py -q -x 2>&1 | tail -2
MODGPU_PART_DIGITS=0 MODGPU_FIND_PATH=2 python -m pytest tests/test_gpu_modset.py tests/test_dropin.py -q -x -m gpu 2>&1 | tail -2
MODGPU_PART_DIGITS=0 MODGPU_TABLE_PATH=bucket python tests/fuzz_gpu.py 29 150 2>&1 | tail -2
echo "== round 6: the table tightened after the dedup kernel's count at every bucketed add into an empty table (70 %, and 95 %: no slack but the fullest bucket's), never, and a table sized at 30 / 95 % of the default"
for kn in "MODGPU_TIGHT_LOAD=70" "MODGPU_TIGHT_LOAD=95" "MODGPU_TIGHT_LOAD=0" "MODGPU_TABLE_LOAD=30" "MODGPU_TABLE_LOAD=95" "MODGPU_BUCKET_R=2048 MODGPU_BUCKET_T=512"; do
  echo "-- $kn"
  env $kn MODGPU_TABLE_PATH=bucket python -m pytest tests/test_gpu_modset.py tests/test_readset.py tests/test_dropin.py -q -x -m gpu 2>&1 | tail -2
  env $kn MODGPU_TABLE_PATH=bucket python tests/fuzz_gpu.py 23 150 2>&1 | tail -2
done
