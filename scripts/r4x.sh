set -o pipefail
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r4x -- python3 $ROOT/bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/prof_r4x.log 2>&1 || exit 1
cd $ROOT
OCF_STEP_SEQ=1 python scripts/step_breakdown.py $OUT/prof_r4x 40 > $OUT/r4x_breakdown.txt 2>&1
rm -rf $OUT/prof_r4x
head -16 $OUT/r4x_breakdown.txt
