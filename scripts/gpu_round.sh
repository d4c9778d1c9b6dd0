#!/bin/bash
# One GPU-box call: kernel micro-benchmark + the parity tests named on the command line.  usage: scripts/gpu_round.sh <tag> [pytest args...]
set -o pipefail
TAG=${1:-x}; shift
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 300 tools/kbench 8 bf16 cl > $OUT/kbench_$TAG.txt 2>&1 || { echo "kbench failed"; tail -5 $OUT/kbench_$TAG.txt; exit 1; }
cat $OUT/kbench_$TAG.txt
if [ $# -gt 0 ]; then
  timeout -k 10 1000 python -m pytest "$@" -x -q 2>&1 | grep -v "MIOpen(HIP): Warning" | tail -25 | tee $OUT/pytest_$TAG.txt
fi
