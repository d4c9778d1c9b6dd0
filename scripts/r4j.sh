set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py tests/test_train_goldens_gpu.py -q -m gpu > $OUT/r4j_models_full.txt 2>&1
grep -v "MIOpen(HIP): Warning" $OUT/r4j_models_full.txt | grep -n "Fatal\|Aborted\|fault\|^E  \|PASSED\|FAILED\|passed\|failed" | head -60
