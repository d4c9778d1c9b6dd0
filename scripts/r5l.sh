set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_models_gpu.py tests/test_runtime_state_gpu.py tests/test_train_goldens_gpu.py -q -x > $OUT/r5l_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5l_tests.txt | tail -4
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/r5l_bench.json 2> $OUT/r5l_bench.err || { tail -20 $OUT/r5l_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5l_bench.json")); print("headline", d["value"], d["ms_per_step"], "unsup", d["unsupervised"]["value"], "cfg3", d["cfg3"]["value"], "frac", d["roofline"]["frac"])
print({k:v for k,v in d["roofline"]["kernels"].items() if "wgrad" in k or "igemm" in k})
PY
