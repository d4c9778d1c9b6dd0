set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python bench.py > $OUT/r4y_bench.json 2> $OUT/r4y_bench.err || { grep -v "MIOpen(HIP)" $OUT/r4y_bench.err | tail -20; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4y_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsupervised']['value'], d['cfg3']['value'], d['config'].get('final_loss'), d['unsupervised'].get('final_loss'), d['cfg3'].get('final_loss'))
PY
timeout -k 10 600 python -m pytest tests/test_train_goldens_gpu.py tests/test_models_gpu.py -q > $OUT/r4y_t2.txt 2>&1
grep -v "MIOpen(HIP): Warning" $OUT/r4y_t2.txt | grep -E "^E  |FAILED|passed|failed|Abort|Fatal" | head -20
