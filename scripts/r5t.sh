set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
for v in 0 2048 8192 32768; do
OCFLOW_WGRAD_STREAM_MAX_PIXELS=$v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cfg3-steps 0 --unsup-steps 0 > $OUT/r5t_bench_$v.json 2> $OUT/r5t_bench_$v.err || { tail -20 $OUT/r5t_bench_$v.err; exit 1; }
python - $v <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r5t_bench_{sys.argv[1]}.json")); print("wgrad stream max pixels", sys.argv[1], d["value"], d["ms_per_step"])
PY
done
