set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
# diagnostic: the order that aborted (goldens first, then the configuration tests), launcher test in-process, HIP error logging on
OCFLOW_TEST_LAUNCHER_INPROC=1 LD_PRELOAD=$PWD/tools/diag/abrt_trace.so timeout -k 10 900 python -m pytest -p no:faulthandler tests/test_train_goldens_gpu.py tests/test_configs_gpu.py -q -x > $OUT/r4z_diag.txt 2>&1
echo "rc $?"
grep -v "MIOpen(HIP): Warning" $OUT/r4z_diag.txt | grep -A70 "native backtrace" | head -90
