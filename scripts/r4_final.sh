# GPU box, end of round 4: the full GPU suite, then the rocprof stats + PMC passes and the step breakdown of the final library
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -q -m gpu > $OUT/r4_final_suite.txt 2>&1
echo "suite rc $?"
grep -v "MIOpen(HIP): Warning" $OUT/r4_final_suite.txt | grep -E "Fatal|Aborted|fault|^E  |FAILED|passed|failed" | head -20
bash scripts/profile_bench.sh r4_final > $OUT/r4_final_profile.log 2>&1
echo "profile rc $?"
bash scripts/r4x.sh > /dev/null 2>&1
head -14 $OUT/r4x_breakdown.txt
