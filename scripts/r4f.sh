set -o pipefail
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r4f -- python3 $ROOT/bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/prof_r4f.log 2>&1 || exit 1
cd $ROOT
python scripts/step_breakdown.py $OUT/prof_r4f 70 > $OUT/r4f_breakdown.txt 2>&1
rm -rf $OUT/prof_r4f
head -50 $OUT/r4f_breakdown.txt
