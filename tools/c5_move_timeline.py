#!/usr/bin/env python3
"""One move of BASELINE configs[4] from a rocprofv3 --kernel-trace CSV: every kernel between two root-noise launches with its duration and the gap in front of it."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(r["Kernel_Name"].replace("void mz::", "").replace("mz::", "").split("(")[0][:48], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
marks = [i for i, n in enumerate(k) if n[0].startswith("sim_root_noise")]
if len(marks) < 4:
    print("no Gumbel-round move in this trace")
    sys.exit(0)
a, b = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
prev = None
tot = {}
for n, s, e in k[a:b]:
    print("%-50s %9.1f us   gap %8.1f us" % (n, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    tot[n] = tot.get(n, 0.0) + (e - s) / 1e3
    prev = e
print("move: %.1f us from the first launch to the next move's; per kernel:" % ((k[b][1] - k[a][1]) / 1e3))
for n, t in sorted(tot.items(), key=lambda x: -x[1]):
    print("  %-50s %9.1f us" % (n, t))
