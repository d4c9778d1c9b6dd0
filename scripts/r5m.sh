set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
echo "== SMR 16" > $OUT/r5m_time_losses.txt
timeout -k 10 300 python scripts/time_losses.py 8 436 1024 >> $OUT/r5m_time_losses.txt 2>&1
python - <<'PY'
from ocflow_amd import build
build.build(force=True, extra_flags=["-DOCF_SMR=8"])
PY
echo "== SMR 8" >> $OUT/r5m_time_losses.txt
timeout -k 10 300 python scripts/time_losses.py 8 436 1024 >> $OUT/r5m_time_losses.txt 2>&1
grep -v amdgpu.ids $OUT/r5m_time_losses.txt
