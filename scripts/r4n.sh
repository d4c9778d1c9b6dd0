OUT=gpurun_out; mkdir -p $OUT
for cfg in "X=1" "OCFLOW_NATIVE_CONV=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 600 python -m pytest "tests/test_train_goldens_gpu.py::test_flownet_occlusion_aware_train_step_matches_reference" "tests/test_models_gpu.py::test_feature_pyramid_pair_equals_two_calls" -q -m gpu 2>&1 | grep -v "MIOpen(HIP)" | grep "AssertionError\|passed\|failed\|^E   *assert" | cut -c1-300
done
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_configs_gpu.py -q -m gpu -k "aligned_concatenation or outside_the_find_db" 2>&1 | grep -v "MIOpen(HIP)" | grep "AssertionError\|passed\|failed\|^E   *assert" | cut -c1-300
