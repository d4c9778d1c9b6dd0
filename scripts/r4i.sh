set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4i_bench1.json 2> $OUT/r4i_bench1.err || { tail -20 $OUT/r4i_bench1.err; exit 1; }
cut -c1-260 $OUT/r4i_bench1.json
OCFLOW_NATIVE_CONV_MIN_PIXELS=32768 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4i_bench0.json 2> $OUT/r4i_bench0.err || { tail -20 $OUT/r4i_bench0.err; exit 1; }
cut -c1-260 $OUT/r4i_bench0.json
OCFLOW_NATIVE_CONV_MIN_PIXELS=8192 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4i_bench2.json 2> $OUT/r4i_bench2.err || { tail -20 $OUT/r4i_bench2.err; exit 1; }
cut -c1-260 $OUT/r4i_bench2.json
