set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest -m gpu tests/test_deconv_gpu.py -q -x > $OUT/r5g_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5g_tests.txt | tail -8
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/bench_deconv.py > $OUT/r5g_bench_deconv.txt 2>&1; tail -19 $OUT/r5g_bench_deconv.txt
OCF_DECONV_CPT=8 timeout -k 10 300 python scripts/bench_deconv.py > $OUT/r5g_bench_deconv_cpt8.txt 2>&1; tail -3 $OUT/r5g_bench_deconv_cpt8.txt
