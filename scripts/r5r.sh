set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_r5r_cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg3 --steps 10 --warmup 3 --preroll 12 --cpu-seconds 0 --unsup-steps 0 > $GRAFT_REPO_ROOT/$OUT/r5r_prof.log 2>&1
cd $GRAFT_REPO_ROOT
OCF_MARKS_PER_STEP=2 OCF_STEP_SEQ=1 python scripts/step_breakdown.py $OUT/prof_r5r_cfg3 90 > $OUT/r5r_cfg3_step_breakdown.txt 2>&1
head -50 $OUT/r5r_cfg3_step_breakdown.txt | cut -c1-160
rm -rf $OUT/prof_r5r_cfg3
