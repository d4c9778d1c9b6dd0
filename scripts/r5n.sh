set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_hip_parity.py tests/test_bench_shapes_gpu.py tests/test_dense_block_gpu.py -q -x > $OUT/r5n_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5n_tests.txt | tail -4
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/r5n_bench.json 2> $OUT/r5n_bench.err || { tail -20 $OUT/r5n_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5n_bench.json")); print("headline", d["value"], d["ms_per_step"], "unsup", d["unsupervised"]["value"], "cfg3", d["cfg3"]["value"], "frac", d["roofline"]["frac"])
print({k:v for k,v in d["roofline"]["kernels"].items() if "smooth" in k or "photo" in k})
PY
