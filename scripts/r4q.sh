set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu -x 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -3
timeout -k 10 500 python scripts/bench_conv.py --levels 2,3 --tw 512 > $OUT/r4q_staged.txt 2>&1 || { tail -5 $OUT/r4q_staged.txt; exit 1; }
tail -24 $OUT/r4q_staged.txt
