#!/bin/bash
# GPU box: rocprofv3 kernel stats of tools/kbench (per-kernel average durations of every hand-written kernel, back-to-back launches)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kbench -- $ROOT/tools/kbench 8 bf16 cl > $OUT/prof_kbench.log 2>&1 || { tail -5 $OUT/prof_kbench.log; exit 1; }
f=$(find $OUT/prof_kbench -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/prof_kbench_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
rm -rf $OUT/prof_kbench
