set -o pipefail
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/r4a_bench.json 2> $OUT/r4a_bench.err || { tail -5 $OUT/r4a_bench.err; exit 1; }
cut -c1-300 $OUT/r4a_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r4a -- python3 $ROOT/bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0 > $OUT/prof_r4a.log 2>&1 || exit 1
cd $ROOT
python scripts/step_breakdown.py $OUT/prof_r4a 90 > $OUT/r4a_breakdown.txt 2>&1
python scripts/conv_inventory.py > $OUT/r4a_conv_inventory.txt 2>&1
rm -rf $OUT/prof_r4a
tail -3 $OUT/r4a_breakdown.txt
