set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python bench.py > $OUT/r4y_bench.json 2> $OUT/r4y_bench.err || { grep -v "MIOpen(HIP)" $OUT/r4y_bench.err | tail -20; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4y_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsupervised']['value'], d['cfg3']['value'], d['roofline']['frac'])
for k,v in d['roofline']['kernels'].items():
    if k.startswith('enc') or k.startswith('conv3x3'):
        print(k, v.get('bound'), v.get('achieved'), v.get('unit', 'GB/s'), v.get('frac'))
PY
