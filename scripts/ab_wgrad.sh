#!/bin/bash
# GPU box: kernel-level comparison + end-to-end A/B of the hand-written weight gradient (OCFLOW_NATIVE_WGRAD=1/0)
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python scripts/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_wgrad.txt
for v in 1 0; do
  OCFLOW_NATIVE_WGRAD=$v timeout -k 10 400 python bench.py --cpu-seconds 0 --unsup-steps 0 2> $OUT/ab_wgrad_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NATIVE_WGRAD=$v', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step loss', d['config']['final_loss'])" || tail -5 $OUT/ab_wgrad_$v.err
done | tee $OUT/ab_wgrad.txt
