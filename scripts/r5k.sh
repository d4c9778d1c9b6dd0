set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python scripts/bench_wgrad3.py --levels 4,5,6 --policies 0,8,2000008,1000008 --rounds 3 > $OUT/r5k_bench_wgrad3_chunks8.txt 2>&1; tail -26 $OUT/r5k_bench_wgrad3_chunks8.txt
