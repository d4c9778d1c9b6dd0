#!/bin/bash
# GPU box: bench.py under the kernel-choice knobs (headline value + the in-model per-launch times of the op named)
set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --unsup-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('pairs/s', d['value'], '| warp_bwd', {s:x['avg_us'] for s,x in k['warp_bwd']['per_shape'].items()}, '| corr_bwd', {s:x['avg_us'] for s,x in k['corr_bwd']['per_shape'].items()})
" || exit 1; }
run OCFLOW_WARP_BWD_POLICY=0
run OCFLOW_WARP_BWD_POLICY=1

