#!/usr/bin/env python3
"""GPU box: what ONE graph-replayed training step of bench.py spends per kernel.  Reads a rocprofv3 kernel trace (CSV), cuts it at the
coarsest-level correlation launch (one per supervised step), keeps the steps whose length is within 10 % of the median (the replays) and
prints, per kernel name, launches per step and microseconds per step, plus category sums (library convolutions by direction, their helper
launches, this repo's kernels, framework glue).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_x -- python3 bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0
  python scripts/step_breakdown.py gpurun_out/prof_x [top_n]"""
import collections
import csv
import glob
import re
import statistics
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 70
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     (int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0))))
rows.sort()
import os
marks = [i for i, r in enumerate(rows) if "corr_fwd_wave<7" in r[2] or "corr_fwd_waveILi7" in r[2]]
# OCF_MARKS_PER_STEP=2: the occlusion-aware unsupervised step runs the network twice (models/model.py:376,384): two coarsest-level
# correlation launches per step
mps = int(os.environ.get("OCF_MARKS_PER_STEP", "1"))
marks = marks[::mps]
steps = [(rows[b][0] - rows[a][0], a, b) for a, b in zip(marks[:-1], marks[1:])]
med = statistics.median(s[0] for s in steps)
rep = [s for s in steps if abs(s[0] - med) < 0.1 * med]
print(f"{len(steps)} steps, median {med / 1e6:.3f} ms, {len(rep)} replayed steps used")


def short(name):
    name = re.sub(r"^void ", "", name)
    if name.startswith("_ZN2ck"):
        m = re.search(r"kernel_grouped_conv_(\w+?)(?:IN|E)", name)
        return "ck::grouped_conv_" + (m.group(1) if m else "?")
    if name.startswith("igemm_") or name.startswith("naive_conv"):
        return name[:24]
    return name[:70]


def category(name):
    if "igemm_fwd" in name or "grouped_conv_fwd" in name or "naive_conv_ab_nonpacked_fwd" in name or "conv_fwd" in name:
        return "lib conv fwd"
    if "igemm_bwd" in name or "grouped_conv_bwd_data" in name or "naive_conv_ab_nonpacked_bwd" in name:
        return "lib conv dgrad"
    if "igemm_wrw" in name or "grouped_conv_bwd_weight" in name or "naive_conv_ab_nonpacked_wrw" in name:
        return "lib conv wgrad"
    if "SubTensorOp" in name or "batched_transpose" in name or "transpose" in name.lower():
        return "lib helpers (fill/cast/transpose)"
    if "ocf" in name:
        if "conv" in name:
            return "ocf conv"
        if "bias_act" in name:
            return "ocf epilogue"
        if "bn" in name:
            return "ocf batchnorm"
        return "ocf hot path + losses"
    if "CatArray" in name:
        return "aten cat"
    if "multi_tensor" in name or "Fused" in name:
        return "aten optimizer / multi-tensor"
    return "aten other"


per = collections.defaultdict(lambda: [0, 0])
cat = collections.defaultdict(lambda: [0, 0])
tot = 0
for _, a, b in rep:
    for s, e, n, g in rows[a:b]:
        k = short(n)
        per[k][0] += 1
        per[k][1] += e - s
        c = category(n)
        cat[c][0] += 1
        cat[c][1] += e - s
        tot += e - s
n = len(rep)
print(f"kernel time per step {tot / n / 1e6:.3f} ms")
for c, (cnt, t) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c:36s} {cnt / n:7.1f} launches {t / n / 1e3:9.1f} us")
print()
for k, (cnt, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / n / 1e3:9.1f} us {cnt / n:7.1f} x  {k}")

# OCF_STEP_SEQ=1: the launches of ONE replayed step in order (name, grid, microseconds) -- which layer a library kernel belongs to is read
# off its neighbours
import os
if os.environ.get("OCF_STEP_SEQ") == "1" and rep:
    _, a, b = rep[len(rep) // 2]
    print("\nlaunch sequence of one replayed step:")
    for s, e, nme, g in rows[a:b]:
        print(f"{(e - s) / 1e3:8.1f} us  grid {g[0]:>9d},{g[1]:>5d},{g[2]:>3d}  {short(nme)[:90]}")
