#!/bin/bash
# GPU box: A/B of the forward cost-volume kernel variants with the tuning library (tools/tuning_lib, built by scripts/build_tuning_lib.sh)
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/tools/tuning_lib:$LD_LIBRARY_PATH
{
for env in "OCF_ROWS_K2_PASSES=2 OCF_WAVE_STRIPS=0" "OCF_ROWS_K2_PASSES=1 OCF_WAVE_STRIPS=1" "OCF_ROWS_K2_PASSES=2 OCF_WAVE_STRIPS=2"; do
  echo "== $env"
  env $env timeout -k 10 200 tools/kbench 8 bf16 cl 2>&1 | grep -E "corr fwd|fused warp|all levels"
done
} | tee $OUT/ab_fwd_variants.txt
