set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/tools/tuning_lib:$LD_LIBRARY_PATH
for v in 0 1 2 3 4; do
echo "== OCF_ROWS_K1_SHAPE=$v"
OCF_ROWS_K1_SHAPE=$v timeout -k 10 200 tools/kbench 8 bf16 cl 2>&1 | grep -E "corr fwd \(plain|tuning" | tail -1 | cut -c1-140
done
