set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $OUT/r5_suite.txt 2>&1
echo "suite rc $?"
grep -v "MIOpen(HIP): Warning" $OUT/r5_suite.txt | grep -E "Fatal|Aborted|fault|^E  |FAILED|passed|failed|native / library" | head -20
