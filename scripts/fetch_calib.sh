#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of tools/fetch_calib (known byte counts) -> gpurun_out/fetch_calibration.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 $ROOT/tools/fetch_calib.cpp -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fc_$c -- /tmp/fetch_calib > /tmp/fc_$c.log 2>&1 || { tail -5 /tmp/fc_$c.log; exit 1; }
done
python3 - <<'PY' > $OUT/fetch_calibration.txt
import csv, glob, collections
bytes_per_kernel = 768 << 20
print("tools/fetch_calib on MI355X: counter * 1024 / streamed bytes (768 MiB per kernel, HBM-resident; second launch of each kernel)")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = glob.glob(f"/tmp/fc_{c}/**/*counter_collection.csv", recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == c:
            vals[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        print(f"  {c:10s} {k:90s} raw {v[-1]:14.1f}  x1024/bytes = {v[-1] * 1024 / bytes_per_kernel:6.3f}")
PY
cat $OUT/fetch_calibration.txt
