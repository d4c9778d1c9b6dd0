set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_conv_igemm_gpu.py tests/test_dense_block_gpu.py -q -x > $OUT/r5i_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5i_tests.txt | tail -8
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python scripts/bench_conv.py --levels 2,3 --ours-only > $OUT/r5i_bench_conv.txt 2>&1; tail -22 $OUT/r5i_bench_conv.txt
