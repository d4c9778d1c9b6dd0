set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_conv_igemm_gpu.py tests/test_hip_parity.py -q -m gpu -x -k "conv or concat or weight or chain or forward or backward or packed or patch or slice or padded or mask" 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -4
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4l_bench1.json 2> $OUT/r4l_bench1.err || { tail -20 $OUT/r4l_bench1.err; exit 1; }
cut -c1-260 $OUT/r4l_bench1.json
OCFLOW_NATIVE_FPN=0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4l_bench0.json 2> $OUT/r4l_bench0.err || { tail -20 $OUT/r4l_bench0.err; exit 1; }
cut -c1-260 $OUT/r4l_bench0.json
OCFLOW_NATIVE_CAT=0 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4l_bench2.json 2> $OUT/r4l_bench2.err || { tail -20 $OUT/r4l_bench2.err; exit 1; }
cut -c1-260 $OUT/r4l_bench2.json
