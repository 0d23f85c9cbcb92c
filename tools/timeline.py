#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV as a per-cycle timeline: kernel durations and the idle gaps between
consecutive kernels, grouped by (previous kernel -> next kernel).  The gaps are where the host (env replay, feature
packing, candidate lists, per-move logic) and launch/completion latency sit.

usage: python tools/timeline.py gpurun_out/trace/**/**_kernel_trace.csv [--skip N]
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"mz::(\w+)", name)
    return m.group(1) if m else name.split("(")[0][-32:]


def main():
    paths = [p for a in sys.argv[1:] if not a.startswith("--") for p in glob.glob(a, recursive=True)]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # drop everything before the `skip`-th select_kernel (warm-up, weight upload)
    sel = [i for i, r in enumerate(rows) if r[2] == "select_kernel"]
    if sel and skip < len(sel):
        rows = rows[sel[skip]:]
    dur = defaultdict(list)
    gap = defaultdict(list)
    for i, (s, e, n) in enumerate(rows):
        dur[n].append(e - s)
        if i:
            gap[(rows[i - 1][2], n)].append(s - rows[i - 1][1])
    sel = [r[0] for r in rows if r[2] == "select_kernel"]
    cyc = [(b - a) for a, b in zip(sel, sel[1:])]
    out = {"cycles": len(cyc), "cycle_us_mean": sum(cyc) / max(1, len(cyc)) / 1e3,
           "cycle_us_median": sorted(cyc)[len(cyc) // 2] / 1e3 if cyc else None,
           "kernel_us": {k: {"n": len(v), "mean": sum(v) / len(v) / 1e3, "per_cycle": sum(v) / max(1, len(cyc)) / 1e3} for k, v in dur.items()},
           "gap_us": {f"{a} -> {b}": {"n": len(v), "mean": sum(v) / len(v) / 1e3, "median": sorted(v)[len(v) // 2] / 1e3,
                                      "per_cycle": sum(v) / max(1, len(cyc)) / 1e3}
                      for (a, b), v in sorted(gap.items(), key=lambda kv: -sum(kv[1])) if len(v) >= 3}}
    out["busy_frac"] = sum(sum(v) for v in dur.values()) / max(1, rows[-1][1] - rows[0][0])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
