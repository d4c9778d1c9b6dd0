set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_conv_igemm_gpu.py -q -m gpu -x 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -3
for p in 512 8704; do
  echo "== policy $p (512 staged, row epilogue; 8704 = +8192 direct 8-byte stores)"
  timeout -k 10 300 python scripts/bench_conv.py --levels 2,3 --tw $p --rounds 4 2>&1 | grep "d 1\|d16\|sum"
done
