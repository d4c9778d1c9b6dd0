OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1100 python -X faulthandler -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py tests/test_train_goldens_gpu.py -v -s -m gpu > $OUT/r4x_models_full.txt 2>&1
echo "rc=$?"
grep -v "MIOpen(HIP): Warning" $OUT/r4x_models_full.txt | grep -n "FAILED\|Fatal\|Abort\|abort\|HIP error\|hipError\|fault\|passed\|failed" | tail -12
L=$(grep -n "Fatal Python error" $OUT/r4x_models_full.txt | head -1 | cut -d: -f1); if [ -n "$L" ]; then sed -n "$((L-30)),$((L+3))p" $OUT/r4x_models_full.txt | grep -v "MIOpen(HIP)" | cut -c1-250; fi
