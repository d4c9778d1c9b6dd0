set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_deconv_gpu.py -q -x > $OUT/r4t_deconv.txt 2>&1
rc=$?
grep -v "MIOpen(HIP): Warning" $OUT/r4t_deconv.txt | tail -5
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $OUT/r4t_bench.json 2> $OUT/r4t_bench.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4t_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsupervised']['value'], d['cfg3']['value'])
PY
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py tests/test_train_gpu.py -q -x > $OUT/r4t_models.txt 2>&1
grep -v "MIOpen(HIP): Warning" $OUT/r4t_models.txt | grep -n "Fatal\|Aborted\|fault\|^E  \|FAILED\|passed\|failed" | head -20
