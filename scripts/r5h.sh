set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_dense_block_gpu.py tests/test_models_gpu.py tests/test_configs_gpu.py -q -x > $OUT/r5h_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5h_tests.txt | tail -12
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --config cfg3 --steps 10 --warmup 3 --preroll 12 --cpu-seconds 0 --unsup-steps 0 > $OUT/r5h_cfg3.json 2> $OUT/r5h_cfg3.err || { tail -20 $OUT/r5h_cfg3.err; exit 1; }
OCFLOW_SHARED_ENCODER=0 timeout -k 10 600 python bench.py --config cfg3 --steps 10 --warmup 3 --preroll 12 --cpu-seconds 0 --unsup-steps 0 > $OUT/r5h_cfg3_noshare.json 2> $OUT/r5h_cfg3_noshare.err || { tail -20 $OUT/r5h_cfg3_noshare.err; exit 1; }
python - <<'PY'
import json
for n in ("","_noshare"):
    d=json.load(open(f"gpurun_out/r5h_cfg3{n}.json")); print(n or "shared", d["value"], d["ms_per_step"], d["config"]["final_loss"])
PY
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/r5h_bench.json 2> $OUT/r5h_bench.err || { tail -20 $OUT/r5h_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5h_bench.json")); print("headline", d["value"], d["ms_per_step"], "unsup", d["unsupervised"]["value"], "cfg3", d["cfg3"]["value"], "frac", d["roofline"]["frac"])
print({k:v for k,v in d["roofline"]["kernels"].items() if "deconv" in k})
PY
timeout -k 10 400 python scripts/bench_conv.py --levels 2,3 --ours-only > $OUT/r5h_bench_conv.txt 2>&1; tail -22 $OUT/r5h_bench_conv.txt
