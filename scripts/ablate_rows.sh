#!/bin/bash
# GPU box: tuning build of the library (-DOCF_TUNING_BUILD) + kbench under the phase-skipping switches of the row kernel.
# Results of the ablated runs are WRONG by construction; only their times are of interest.  Output: gpurun_out/ablate_rows.txt
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python -c "from ocflow_amd import build; build.build(force=True, extra_flags=('-DOCF_TUNING_BUILD',))" || exit 1
for pad in 0 512 1024 2048 4096; do
  echo "== OCF_LDS_PAD=$pad"
  OCF_LDS_PAD=$pad timeout -k 10 120 tools/kbench 8 bf16 cl 2>&1 | grep -E "tuning|variant auto [12] "
done | tee $OUT/ablate_rows.txt
for ab in 1 2 4 8 15; do
  echo "== OCF_ABLATE=$ab (1: no MFMA, 2: no band stores, 4: no copy-out, 8: no global loads)"
  OCF_ABLATE=$ab timeout -k 10 120 tools/kbench 8 bf16 cl 2>&1 | grep -E "variant auto [12] " 
done | tee -a $OUT/ablate_rows.txt
python -c "from ocflow_amd import build; build.build(force=True)"
