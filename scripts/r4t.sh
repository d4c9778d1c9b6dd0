set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_deconv_gpu.py -q -x > $OUT/r4t_deconv.txt 2>&1
rc=$?
grep -v "MIOpen(HIP): Warning" $OUT/r4t_deconv.txt | tail -3
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/bench_deconv.py 2>&1 | grep -v "MIOpen(HIP): Warning" | tee $OUT/r4t_bench_deconv.txt
