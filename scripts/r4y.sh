set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_conv_igemm_gpu.py -q -k "c3_first" > $OUT/r4y_t1.txt 2>&1
rc=$?
grep -v "MIOpen(HIP): Warning" $OUT/r4y_t1.txt | grep -E "^E  |FAILED|passed|failed" | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $OUT/r4y_bench.json 2> $OUT/r4y_bench.err || { grep -v "MIOpen(HIP)" $OUT/r4y_bench.err | tail -20; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4y_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsupervised']['value'], d['cfg3']['value'], d['config'].get('final_loss'))
k=d['roofline']['kernels']
for n in ('enc3_wgrad','enc3_conv_s2','enc16_wgrad'):
    if n in k: print(n, k[n].get('avg_launch_us'), k[n].get('achieved'))
PY
