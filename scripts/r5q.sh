set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 300 python scripts/bench_deconv4.py > $OUT/r5q_micro.txt 2>&1 || { tail $OUT/r5q_micro.txt; exit 1; }
cat $OUT/r5q_micro.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/q -o q -- python $GRAFT_REPO_ROOT/scripts/bench_deconv4.py --ours-only > /dev/null 2>&1
f=$(find /tmp/q -name "*kernel_stats.csv" | head -1); grep -i "deconv4x4\|Name" $f | cut -c1-200
