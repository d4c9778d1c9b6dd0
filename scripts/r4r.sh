set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 500 python scripts/bench_wgrad3.py --levels 2,3 --policies 0,800000 > $OUT/r4r_wgrad.txt 2>&1; tail -19 $OUT/r4r_wgrad.txt
