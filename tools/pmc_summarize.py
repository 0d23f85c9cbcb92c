#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) per kernel: mean counter value per dispatch."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(dirs, out):
    res = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    short = name.split("(")[0].replace("void ", "")
                    res[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    summary = {k: {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in res.items()}
    with open(out, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for k, cs in sorted(summary.items()):
        print(k, {c: round(v["mean"], 1) for c, v in cs.items()})


if __name__ == "__main__":
    main(sys.argv[2:], sys.argv[1])
