set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu -x 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -3
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4s_bench$i.json 2> $OUT/r4s_bench$i.err || { tail -20 $OUT/r4s_bench$i.err; exit 1; }
cut -c1-200 $OUT/r4s_bench$i.json
done
