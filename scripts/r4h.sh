set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu -x 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -4
timeout -k 10 500 python scripts/bench_conv.py --levels 4,5,6 --tw 256 > $OUT/r4h_direct.txt 2>&1 || { tail -5 $OUT/r4h_direct.txt; exit 1; }
tail -30 $OUT/r4h_direct.txt
