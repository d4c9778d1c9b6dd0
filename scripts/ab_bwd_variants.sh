#!/bin/bash
# GPU box: A/B of the correlation-backward variants with the tuning library (tools/tuning_lib, built by scripts/build_tuning_lib.sh):
# passes side by side in one launch (default; _PHASE=0: built from corr_bwd_pass instead of the phase functions), one after the other in
# a workgroup, two launches, two launches with the per-lane gradient loads.  (The pipelined variant of profiles/r3_h_* was removed.)
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/tools/tuning_lib:$LD_LIBRARY_PATH
{
for env in ${AB_ENVS:-"OCF_BWD_PAR=1" "OCF_BWD_PAR=1 OCF_BWD_PAR_PHASE=0" "OCF_BWD_PAR=0 OCF_BWD_FUSED=1" "OCF_BWD_PAR=0 OCF_BWD_FUSED=0" "OCF_BWD_PAR=0 OCF_BWD_FUSED=0 OCF_BWD_G1_LDS=0"}; do
  echo "== $env"
  env $env timeout -k 10 200 tools/kbench 8 bf16 cl 2>&1 | grep -E "corr_bwd\["
done
} | tee $OUT/ab_bwd_variants.txt
