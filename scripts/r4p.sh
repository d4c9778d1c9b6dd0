set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "concat" 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -3
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4p_bench1.json 2> $OUT/r4p_bench1.err || { tail -20 $OUT/r4p_bench1.err; exit 1; }
cut -c1-260 $OUT/r4p_bench1.json
