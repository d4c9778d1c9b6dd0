set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python bench.py > $OUT/r4o_bench_default.json 2> $OUT/r4o_bench_default.err || { tail -20 $OUT/r4o_bench_default.err; exit 1; }
cut -c1-400 $OUT/r4o_bench_default.json
bash scripts/profile_bench.sh r4o > $OUT/r4o_profile.log 2>&1 || { tail -20 $OUT/r4o_profile.log; exit 1; }
tail -5 $OUT/r4o_profile.log
