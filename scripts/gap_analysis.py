#!/usr/bin/env python3
"""GPU box: how much of a replayed step is the GPU idle?  Reads a rocprofv3 kernel trace (CSV) of bench.py, takes the coarsest-level
correlation launch (one per supervised step) as the step marker and, for every step whose length is within 10 % of the median (the
hipGraph replays), reports step length, summed kernel time, idle time and the number of kernels.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python3 bench.py --steps 10 --warmup 3 --preroll 22 --cpu-seconds 0 --unsup-steps 0 --cfg3-steps 0
  python scripts/gap_analysis.py gpurun_out/prof_gap"""
import csv
import glob
import statistics
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "corr_fwd_wave<7" in r[2] or "corr_fwd_waveILi7" in r[2]]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    length = rows[b][0] - rows[a][0]
    busy = sum(e - s for s, e, _ in seg)
    # union of intervals (kernels of one stream do not overlap, but be safe)
    cover, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += (cur_e - cur_s) if cur_e is not None else 0
    steps.append((length, busy, cover, len(seg)))
med = statistics.median(s[0] for s in steps)
rep = [s for s in steps if abs(s[0] - med) < 0.1 * med]
print(f"{len(steps)} steps, median length {med / 1e6:.3f} ms; {len(rep)} within 10 % of it")
L = statistics.mean(s[0] for s in rep) / 1e6
Bz = statistics.mean(s[1] for s in rep) / 1e6
Cv = statistics.mean(s[2] for s in rep) / 1e6
N = statistics.mean(s[3] for s in rep)
print(f"step {L:.3f} ms | kernel time {Bz:.3f} ms | covered {Cv:.3f} ms | idle {L - Cv:.3f} ms ({100 * (L - Cv) / L:.1f} %) | {N:.0f} kernels | "
      f"{1e3 * (L - Cv) / N:.2f} us idle per kernel")
# the biggest gaps
seg = rows[marks[len(marks) // 2]:marks[len(marks) // 2 + 1]]
gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1))[::-1][:12]
for g, a, b in gaps:
    print(f"  gap {g:7.1f} us after {a} -> {b}")
