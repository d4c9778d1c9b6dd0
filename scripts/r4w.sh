set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
for v in auto 0; do
OCFLOW_PAD_PWC=$v python bench.py --model pwoc --height 384 --width 1024 --steps 10 --warmup 3 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4w_pwoc_$v.json 2> $OUT/r4w_pwoc_$v.err || { tail -20 $OUT/r4w_pwoc_$v.err; exit 1; }
cut -c1-200 $OUT/r4w_pwoc_$v.json
done
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py tests/test_train_goldens_gpu.py -q -m gpu 2>&1 | grep -v "MIOpen(HIP)" | tail -6
