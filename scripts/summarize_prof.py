#!/usr/bin/env python3
"""Condense rocprofv3 output of scripts/profile_bench.sh into small, committable summaries.

  python scripts/summarize_prof.py gpurun_out/prof_<tag>  ->  <dir>_summary.json + <dir>_kernel_stats.csv (top 60 rows)

Per hand-written kernel (name contains 'ocf::'): calls, average/min/max duration from the kernel trace, grouped by TOTAL grid size
(x * y * z work-items = pyramid level; the counter CSV of rocprofv3 only has the total, the trace CSV has the three components --
round 3 keyed the trace by x alone and attached the counters of multi-dimensional grids to the wrong level); HBM traffic per launch from
the PMC passes, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request on wide coalesced
reads -> x2 for 16-byte-per-lane streams; the factor for 8- and 12-byte per-lane loads is measured by tools/fetch_calib.cpp, see
profiles/r4_fetch_calibration.txt; values are in KiB... the counter unit is bytes/1024).
The raw traces (tens of MB) are deleted unless --keep."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    """a readable kernel name.  rocprofv3 leaves names whose template arguments include __bf16 (`DF16b`) MANGLED (_ZN3ocf...): those are
    rebuilt from the length-prefixed identifiers of the nested name + the raw template-argument string."""
    if name.startswith("_ZN"):
        import re
        i, parts = 3, []
        while i < len(name) and name[i].isdigit():
            m = re.match(r"\d+", name[i:])
            n = int(m.group(0))
            i += len(m.group(0))
            parts.append(name[i:i + n])
            i += n
        targs = ""
        if i < len(name) and name[i] == "I":  # template arguments up to the matching 'E' of the nested name
            j = name.find("EEv", i)
            targs = "<" + name[i + 1:(j if j > 0 else i + 24)].replace("DF16b", "bf16,").replace("Li", "").replace("Lb", "").replace("E", ",").strip(",") + ">"
        return ("::".join(parts) + targs)[:90]
    n = name.split("(")[0]
    n = n.replace("void ", "")
    return n[:90]


def ours(name):
    return "ocf::" in name or name.startswith("_ZN3ocf")


def trace_groups(path):
    groups = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if not ours(name):
                continue
            if row.get("Grid_Size_X"):
                grid = str(int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y") or 1) * int(row.get("Grid_Size_Z") or 1))
            else:
                grid = row.get("Grid_Size") or "?"
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            groups[(short(name), grid)].append(dur)
    return groups


def counter_groups(path):
    groups = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if not ours(name):
                continue
            if row.get("Grid_Size"):
                grid = row["Grid_Size"]
            elif row.get("Grid_Size_X"):
                grid = str(int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y") or 1) * int(row.get("Grid_Size_Z") or 1))
            else:
                grid = "?"
            groups[(short(name), grid, row.get("Counter_Name", "?"))].append(float(row.get("Counter_Value", 0)))
    return groups


def main():
    base = sys.argv[1].rstrip("/")
    keep = "--keep" in sys.argv
    out = {"source": os.path.basename(base), "kernels": {}}
    tr = find(base, "*kernel_trace.csv")
    if tr:
        for (name, grid), durs in sorted(trace_groups(tr).items()):
            k = out["kernels"].setdefault(name, {})
            k[f"grid{grid}"] = {"calls": len(durs), "avg_us": round(sum(durs) / len(durs), 2), "min_us": round(min(durs), 2),
                                "max_us": round(max(durs), 2)}
    for suffix, cname in (("_fetch", "FETCH_SIZE"), ("_write", "WRITE_SIZE")):
        cc = find(base + suffix, "*counter_collection.csv")
        if not cc:
            continue
        for (name, grid, counter), vals in sorted(counter_groups(cc).items()):
            k = out["kernels"].setdefault(name, {}).setdefault(f"grid{grid}", {})
            avg = sum(vals) / len(vals)
            k[f"{counter}_raw_avg"] = round(avg, 2)
            # KiB -> bytes; gfx950: FETCH_SIZE reads half of the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM)
            k[f"{counter}_bytes_corrected"] = int(avg * 1024 * (2 if counter == "FETCH_SIZE" else 1))
    with open(base + "_summary.json", "w") as f:
        json.dump(out, f, indent=1)
    st = find(base, "*kernel_stats.csv")
    if st:
        with open(st) as f, open(base + "_kernel_stats.csv", "w") as g:
            for i, line in enumerate(f):
                # keep every row, shorten the (templated) kernel names: '"name",calls,total,avg,pct,min,max,std'
                line = line.rstrip("\n")
                if line.startswith('"') and '",' in line:
                    name, rest = line[1:].rsplit('",', 1)
                    line = '"' + name[:160].replace('"', "'") + '",' + rest
                g.write(line + "\n")
    if not keep:
        for d in (base, base + "_fetch", base + "_write"):
            for p in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
                os.remove(p)
    print(json.dumps(out, indent=1)[:6000])


if __name__ == "__main__":
    main()
