set -o pipefail
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_bench_shapes_gpu.py tests/test_deconv_gpu.py -q > $OUT/r4w_tests.txt 2>&1
rc=$?
grep -v "MIOpen(HIP): Warning" $OUT/r4w_tests.txt | grep -E "FAILED|passed|failed" | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $OUT/r4w_bench.json 2> $OUT/r4w_bench.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4w_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['unsupervised']['value'], d['cfg3']['value'], d['roofline'].get('frac'))
for k in d['roofline'].get('kernels', []):
    print(k)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r4w -- python3 $ROOT/bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/prof_r4w.log 2>&1 || exit 1
cd $ROOT
python scripts/step_breakdown.py $OUT/prof_r4w 70 > $OUT/r4w_breakdown.txt 2>&1
rm -rf $OUT/prof_r4w
head -80 $OUT/r4w_breakdown.txt
