#!/bin/bash
# GPU box: does another MIOpen weight-gradient solver family avoid the workspace fill + cast helper launches?  (find mode re-tunes)
OUT=gpurun_out; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" OCFLOW_NATIVE_WGRAD=0 OCFLOW_MIOPEN_FIND=1 MIOPEN_USER_DB_PATH=/tmp/mi_$name/config MIOPEN_CUSTOM_CACHE_DIR=/tmp/mi_$name/cache \
    timeout -k 10 500 python bench.py --cpu-seconds 0 --unsup-steps 0 2> $OUT/miopen_exp_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')" || tail -3 $OUT/miopen_exp_$name.err
}
mkdir -p /tmp/mi_base/config /tmp/mi_base/cache /tmp/mi_noasm/config /tmp/mi_noasm/cache
run base X=1
run noasm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
