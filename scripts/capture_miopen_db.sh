#!/bin/bash
# GPU box: regenerate the MIOpen find-db / kernel cache the repo ships (ocflow_amd/miopen_db) with a find-mode run of the benchmark
# (both model legs) and of configs[2]; the result lands in gpurun_out/miopen_db (copy it over ocflow_amd/miopen_db and commit).
set -o pipefail
rm -rf gpurun_out/miopen_db
OCFLOW_MIOPEN_FIND=1 OCFLOW_MIOPEN_CAPTURE=$PWD/gpurun_out/miopen_db timeout -k 10 900 python bench.py --steps 5 --warmup 3 > gpurun_out/capture_bench.json 2> gpurun_out/capture_bench.err || { tail -5 gpurun_out/capture_bench.err; exit 1; }
echo "captured (default workload): $(python -c "import json;d=json.loads(open('gpurun_out/capture_bench.json').read().strip().splitlines()[-1]);print(d['value'], d['unsupervised']['value'])")"
OCFLOW_MIOPEN_FIND=1 OCFLOW_MIOPEN_CAPTURE=$PWD/gpurun_out/miopen_db timeout -k 10 600 python bench.py --config cfg3 --steps 3 --warmup 2 --unsup-steps 0 > gpurun_out/capture_cfg3.json 2> gpurun_out/capture_cfg3.err || { tail -5 gpurun_out/capture_cfg3.err; exit 1; }
du -sh gpurun_out/miopen_db; ls -la gpurun_out/miopen_db/*
