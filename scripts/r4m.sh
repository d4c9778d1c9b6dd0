set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1150 python -m pytest tests -q -m gpu > $OUT/r4m_full.txt 2>&1
grep -v "MIOpen(HIP): Warning" $OUT/r4m_full.txt | grep -n "Fatal\|Aborted\|fault\|^E  \|FAILED\|passed\|failed" | head -40
