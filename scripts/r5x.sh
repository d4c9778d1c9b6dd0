set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_r5x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --cfg3-steps 0 --unsup-steps 0 > $GRAFT_REPO_ROOT/$OUT/r5x_prof.log 2>&1
cd $GRAFT_REPO_ROOT
OCF_STEP_SEQ=1 python scripts/step_breakdown.py $OUT/prof_r5x 90 > $OUT/r5x_step_breakdown.txt 2>&1
head -14 $OUT/r5x_step_breakdown.txt | cut -c1-160
grep "s2::" $OUT/r5x_step_breakdown.txt | cut -c1-150
rm -rf $OUT/prof_r5x
