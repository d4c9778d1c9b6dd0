set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
for v in auto 0; do
OCFLOW_PAD_PWC=$v python bench.py --config cfg3 --steps 10 --warmup 3 --cpu-seconds 0 > $OUT/r4v_cfg3_$v.json 2> $OUT/r4v_cfg3_$v.err || { tail -20 $OUT/r4v_cfg3_$v.err; exit 1; }
cut -c1-300 $OUT/r4v_cfg3_$v.json
done
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py tests/test_models_gpu.py -q -m gpu -k "pwc or config2 or pwoc" 2>&1 | grep -v "MIOpen(HIP)" | tail -5
