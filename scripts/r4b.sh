set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -30 | tee $OUT/r4b_pytest.txt
timeout -k 10 500 python scripts/bench_conv.py --levels 2,3,4,5,6 > $OUT/r4b_bench_conv.txt 2>&1; tail -60 $OUT/r4b_bench_conv.txt
