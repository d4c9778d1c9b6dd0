#!/bin/bash
# GPU box: tuning build; the matrix-core corr backward with phases switched off (bit 0 = grad_out operand loads hit one address,
# 1 = other-operand staging loads hit one address, 2 = no MFMA loop, 3 = no stores).  Timings only (results are wrong by design).
set -o pipefail
python -c "from ocflow_amd import build; build.build(force=True, extra_flags=('-DOCF_TUNING_BUILD',))" || exit 1
for t in 0 1 2 4 8 15 16; do echo "== ablate=$t"; OCF_ABLATE_BWD=$t timeout -k 10 120 tools/kbench 8 bf16 cl 2>&1 | grep -E "corr_bwd: vector" | tail -2 || exit 1; done
