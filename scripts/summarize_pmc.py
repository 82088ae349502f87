#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: [0.0, 0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
            acc[k][0] += float(row.get("Counter_Value", 0) or 0)
            acc[k][1] += 1
    print("==", os.path.relpath(f, root))
    for (kern, ctr), (s, n) in sorted(acc.items()):
        print(f"{kern:60s} {ctr:28s} mean={s / n:.6g}  n={n}")
