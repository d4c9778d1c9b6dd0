set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_bench_shapes_gpu.py -q -k "warp" > $OUT/r4v_warp.txt 2>&1
rc=$?
grep -v "MIOpen(HIP): Warning" $OUT/r4v_warp.txt | grep -E "FAILED|passed|failed" | head -30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 tools/kbench 8 bf16 cl > $OUT/r4v_kbench.txt 2>&1 || { tail -5 $OUT/r4v_kbench.txt; exit 1; }
grep -E "^ *[0-9]+x" $OUT/r4v_kbench.txt
