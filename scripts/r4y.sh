set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/tools/tuning_lib:$LD_LIBRARY_PATH
for cap in 768 600 512 448 384; do
echo "== OCF_ROWS_CAP=$cap"
OCF_ROWS_CAP=$cap timeout -k 10 200 tools/kbench 8 bf16 cl 2>&1 | grep -E "corr fwd \(plain" | tail -2 | cut -c1-120
done
