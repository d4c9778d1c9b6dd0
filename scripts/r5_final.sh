# GPU box, end of round 5: the driver's bench command, the rocprof stats + PMC passes + headline step breakdown, the configs[2] step breakdown
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/r5_final_bench.json 2> $OUT/r5_final_bench.err || { tail -20 $OUT/r5_final_bench.err; exit 1; }
python - <<'PY'
import json
s=open("gpurun_out/r5_final_bench.json").read().strip().splitlines()[-1]
d=json.loads(s); print("line bytes", len(s), "headline", d["value"], d["ms_per_step"], "unsup", d["unsupervised"]["value"], "cfg3", d["cfg3"].get("value"), "frac", d["roofline"]["frac"], "fused", d["roofline"].get("fused_op",{}).get("frac"))
PY
cp profiles/bench_kernels_latest.json $OUT/r5_final_bench_kernels.json 2>/dev/null
bash scripts/r5_profile.sh r5_final || exit 1
bash scripts/r5r.sh > /dev/null && cp $OUT/r5r_cfg3_step_breakdown.txt $OUT/r5_cfg3_step_breakdown_final.txt
head -12 $OUT/r5_cfg3_step_breakdown_final.txt
