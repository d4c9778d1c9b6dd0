set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -m pytest -m gpu tests/test_dense_block_gpu.py tests/test_deconv_gpu.py tests/test_abi.py -q -x > $OUT/r5f_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; grep -v "Warning" $OUT/r5f_tests.txt | tail -25
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/bench_deconv.py > $OUT/r5f_bench_deconv.txt 2>&1; tail -20 $OUT/r5f_bench_deconv.txt
timeout -k 10 600 python bench.py --config cfg3 --steps 10 --warmup 3 --preroll 12 --cpu-seconds 0 --unsup-steps 0 > $OUT/r5e_cfg3_dense.json 2> $OUT/r5e_cfg3_dense.err || { tail -20 $OUT/r5e_cfg3_dense.err; exit 1; }
OCFLOW_NATIVE_DENSE=0 timeout -k 10 600 python bench.py --config cfg3 --steps 10 --warmup 3 --preroll 12 --cpu-seconds 0 --unsup-steps 0 > $OUT/r5e_cfg3_nodense.json 2> $OUT/r5e_cfg3_nodense.err || { tail -20 $OUT/r5e_cfg3_nodense.err; exit 1; }
python - <<'PY'
import json
for n in ("dense","nodense"):
    d=json.load(open(f"gpurun_out/r5e_cfg3_{n}.json")); print(n, d["value"], d["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_r5e_cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg3 --steps 10 --warmup 3 --preroll 12 --cpu-seconds 0 --unsup-steps 0 > $GRAFT_REPO_ROOT/$OUT/r5e_prof.log 2>&1
cd $GRAFT_REPO_ROOT
OCF_MARKS_PER_STEP=2 OCF_STEP_SEQ=1 python scripts/step_breakdown.py $OUT/prof_r5e_cfg3 80 > $OUT/r5e_cfg3_step_breakdown.txt 2>&1
head -60 $OUT/r5e_cfg3_step_breakdown.txt
rm -rf $OUT/prof_r5e_cfg3
