set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_models_gpu.py tests/test_conv_igemm_gpu.py tests/test_configs_gpu.py -q -x > $OUT/r5u_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5u_tests.txt | tail -12
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
OCFLOW_PAD_ENCODER=$v timeout -k 10 400 python bench.py --config cfg3 --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/r5u_cfg3_$v.json 2> $OUT/r5u_cfg3_$v.err || { tail -20 $OUT/r5u_cfg3_$v.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r5u_cfg3_{sys.argv[1]}.json")); print("pad encoder", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
