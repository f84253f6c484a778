#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.

usage: pmc_summary.py <dir-or-csv> [<dir-or-csv> ...] > summary.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    files = []
    for a in sys.argv[1:]:
        if os.path.isdir(a):
            files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(a)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?").split("(")[0]
                ctr = row.get("Counter_Name")
                val = float(row.get("Counter_Value", 0) or 0)
                cell = acc[name][ctr]
                cell[0] += val
                cell[1] += 1
    out = {k: {c: {"mean_per_dispatch": v[0] / max(v[1], 1), "dispatches": v[1], "total": v[0]}
               for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
