# GPU box, round 5: rocprof stats + PMC passes of the bench command (scripts/profile_bench.sh), then the step breakdown of the headline step
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r5_final}
OCF_PROF_KEEP=1 bash scripts/profile_bench.sh $TAG > $OUT/${TAG}_profile.log 2>&1
echo "profile rc $?"; tail -3 $OUT/${TAG}_profile.log
OCF_STEP_SEQ=1 python3 scripts/step_breakdown.py $OUT/prof_${TAG} 70 > $OUT/${TAG}_step_breakdown.txt 2>&1
head -16 $OUT/${TAG}_step_breakdown.txt
rm -rf $OUT/prof_${TAG} $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write
ls $OUT | grep $TAG
