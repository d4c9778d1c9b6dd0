set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest -m gpu tests/test_runtime_state_gpu.py tests/test_train_goldens_gpu.py -q -x > $OUT/r5s_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5s_tests.txt | tail -12
[ $rc -eq 0 ] || exit 1
for v in 1 0; do
OCFLOW_WGRAD_STREAM=$v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cfg3-steps 0 --unsup-steps 0 > $OUT/r5s_bench_$v.json 2> $OUT/r5s_bench_$v.err || { tail -20 $OUT/r5s_bench_$v.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r5s_bench_{sys.argv[1]}.json")); print("wgrad stream", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
