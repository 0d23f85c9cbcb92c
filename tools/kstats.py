#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: short kernel name, calls, average us, share."""
import csv
import glob
import re
import sys

for path in [p for a in sys.argv[1:] for p in glob.glob(a, recursive=True)]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+)(<[^>]*>)?\(", r["Name"])
        print(f"{(m.group(1) if m else r['Name'])[:28]:28s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:8.1f}  {float(r['Percentage']):6.2f}%")
