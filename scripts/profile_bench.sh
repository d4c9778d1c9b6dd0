#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats of the bench command + two PMC passes (HBM traffic counters).
# usage: scripts/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>{,_fetch,_write}/
set -o pipefail
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "[profile] stats pass"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -- python3 $ROOT/bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 4 --cfg3-steps 0 > $OUT/prof_${TAG}.log 2>&1 || exit 1
grep '"metric"' $OUT/prof_${TAG}.log | cut -c1-200
echo "[profile] FETCH_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --preroll 1 --cpu-seconds 0 --unsup-steps 2 --cfg3-steps 0 > $OUT/prof_${TAG}_fetch.log 2>&1 || exit 1
echo "[profile] WRITE_SIZE pass"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --preroll 1 --cpu-seconds 0 --unsup-steps 2 --cfg3-steps 0 > $OUT/prof_${TAG}_write.log 2>&1 || exit 1
true
python3 $ROOT/scripts/summarize_prof.py $OUT/prof_${TAG} ${OCF_PROF_KEEP:+--keep} > $OUT/prof_${TAG}_summary.txt 2>&1
rm -f $OUT/prof_${TAG}*.log.bak
ls -la $OUT | head -30
