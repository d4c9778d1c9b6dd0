set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
OCF_DIAG_PYTEST_FLAGS="" bash scripts/r5_abrt_diag.sh > $OUT/r5d_diag_stdout.txt 2>&1 || { tail -60 $OUT/r5d_diag_stdout.txt; exit 1; }
tail -5 $OUT/r5d_diag_stdout.txt
timeout -k 10 900 python -m pytest -m gpu tests/test_runtime_state_gpu.py tests/test_conv_igemm_gpu.py tests/test_configs_gpu.py::test_config1_flowoccnet_bf16_native_convolutions_gap_to_the_reference -q -x -s > $OUT/r5d_new_tests.txt 2>&1
echo "rc $?"; grep -v "Warning" $OUT/r5d_new_tests.txt | tail -30
