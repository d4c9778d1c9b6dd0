set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu -x 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -5
timeout -k 10 500 python scripts/bench_wgrad3.py --levels 2,3 --policies 0,32,24 > $OUT/r4e_wgrad.txt 2>&1; head -30 $OUT/r4e_wgrad.txt
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4e_bench1.json 2> $OUT/r4e_bench1.err || { tail -20 $OUT/r4e_bench1.err; exit 1; }
cut -c1-260 $OUT/r4e_bench1.json
OCFLOW_NATIVE_WGRAD3=0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4e_bench0.json 2> $OUT/r4e_bench0.err || { tail -20 $OUT/r4e_bench0.err; exit 1; }
cut -c1-260 $OUT/r4e_bench0.json
