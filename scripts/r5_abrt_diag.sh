set -o pipefail
# diagnostic: the order that aborted in round 4 (train goldens first, then the configuration tests) with the launcher test in-process.
# Native + Python traces go to files of their own (tools/diag/abrt_trace.c, tests/conftest.py), stderr/stdout uncaptured (-s).
OUT=gpurun_out; mkdir -p $OUT
gcc -shared -fPIC -O1 -o tools/diag/abrt_trace.so tools/diag/abrt_trace.c || exit 1
rm -f $OUT/r5_abrt_trace.txt $OUT/r5_abrt_trace.txt.python.txt
export LIBC_FATAL_STDERR_=1 OCF_ABRT_TRACE_FILE=$PWD/$OUT/r5_abrt_trace.txt
ulimit -c 0
LD_PRELOAD=$PWD/tools/diag/abrt_trace.so timeout -k 10 900 python -m pytest -p no:faulthandler ${OCF_DIAG_PYTEST_FLAGS--s} -m gpu tests/test_train_goldens_gpu.py tests/test_configs_gpu.py -q -x > $OUT/r5_abrt_diag.txt 2>&1
echo "rc $?"
grep -v "MIOpen(HIP): Warning" $OUT/r5_abrt_diag.txt | tail -40
echo ---- native; head -60 $OUT/r5_abrt_trace.txt 2>/dev/null
echo ---- python; head -80 $OUT/r5_abrt_trace.txt.python.txt 2>/dev/null
