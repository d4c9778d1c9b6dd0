#!/bin/bash
# GPU box: A/B of the correlation-backward variants with the tuning library (tools/tuning_lib, built by scripts/build_tuning_lib.sh)
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/tools/tuning_lib:$LD_LIBRARY_PATH
{
for env in ${AB_ENVS:-"OCF_BWD_PIPE=1 OCF_BWD_PAR=0" "OCF_BWD_PIPE=0 OCF_BWD_PAR=0" "OCF_BWD_PIPE=1 OCF_BWD_PAR=0 OCF_BWD_PIPE_SLOTS=256" "OCF_BWD_PIPE=1 OCF_BWD_PAR=0 OCF_BWD_PIPE_SLOTS=1024" "OCF_BWD_PIPE=0 OCF_BWD_PAR=0" "OCF_BWD_PAR=1" "OCF_BWD_PIPE=0 OCF_BWD_PAR=0 OCF_BWD_FUSED=0"}; do
  echo "== $env"
  env $env timeout -k 10 200 tools/kbench 8 bf16 cl 2>&1 | grep -E "corr_bwd\["
done
} | tee $OUT/ab_bwd_variants.txt
