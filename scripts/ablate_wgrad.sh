#!/bin/bash
# GPU box: tuning build; the LDS-DMA weight-gradient kernel with its phases switched off (1 = no atomic epilogue, 2 = no MFMA loop)
# and with forced split-K chunk counts.
set -o pipefail
python -c "from ocflow_amd import build; build.build(force=True, extra_flags=('-DOCF_TUNING_BUILD',))" || exit 1
for t in 0 1 2 3; do echo "== tune=$t"; OCF_WGRAD_TUNE=$t timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -E "128->128 109x256 d 1|96-> 64 109x256 d 1 " || exit 1; done
for c in 16 32 64 128 256; do echo "== chunks=$c"; OCF_WGRAD_CHUNKS=$c timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -E "128->128 109x256 d 1|96-> 64 109x256 d 1 " || exit 1; done
for c in 32 64 128; do echo "== chunks=$c tune=1"; OCF_WGRAD_TUNE=1 OCF_WGRAD_CHUNKS=$c timeout -k 10 120 python scripts/bench_wgrad.py 2>&1 | grep -E "128->128 109x256 d 1|96-> 64 109x256 d 1 " || exit 1; done
