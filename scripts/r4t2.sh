set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_train_goldens_gpu.py tests/test_models_gpu.py -q > $OUT/r4t_models.txt 2>&1
grep -v "MIOpen(HIP): Warning" $OUT/r4t_models.txt | grep -n "Fatal\|Aborted\|fault\|^E  \|FAILED\|passed\|failed" | head -20
