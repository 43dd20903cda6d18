#!/bin/bash
# GPU suite under every tuning-aid variant of the kernels (run on the GPU box: gpurun -- bash tools/gpu_check_variants.sh)
set -o pipefail
for e in MOBODY_X=0 MOBODY_TILE_ROWS=64 MOBODY_DYN_TILE_ROWS=32 MOBODY_FWD_SHAPE=8 MOBODY_NO_FWD_PAIR=1 MOBODY_MERGE_ACTOR_Q=0; do
  echo "== $e"
  env $e timeout -k 10 400 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_dp.py 2>&1 | tail -1 || exit 1
done
