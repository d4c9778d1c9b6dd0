set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_models_gpu.py tests/test_train_goldens_gpu.py -q -x > $OUT/r5o_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5o_tests.txt | tail -6
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cfg3-steps 0 > $OUT/r5o_bench.json 2> $OUT/r5o_bench.err || { tail -20 $OUT/r5o_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5o_bench.json")); print("headline", d["value"], d["ms_per_step"], "unsup", d["unsupervised"]["value"], d["unsupervised"]["ms_per_step"], d["unsupervised"]["final_loss"], "fused_op", d["roofline"]["fused_op"]["frac"])
PY
