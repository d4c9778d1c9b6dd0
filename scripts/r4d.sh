set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu -x -k "weight" 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -15 | tee $OUT/r4d_pytest.txt
timeout -k 10 500 python scripts/bench_wgrad3.py --levels 2,3 --policies 0,64,48,1000,1032 > $OUT/r4d_bench_wgrad.txt 2>&1; tail -40 $OUT/r4d_bench_wgrad.txt
