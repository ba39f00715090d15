#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: per-kernel average of each counter per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

ALL = "--all" in sys.argv
for d in [a for a in sys.argv[1:] if a != "--all"]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", "nan")))
        print("==", f)
        for k, cs in acc.items():
            if not ALL and "kf_" not in k:
                continue
            if "fk::" not in k and "kf_" not in k:
                continue
            for c, vals in cs.items():
                print(f"{k[:90]:90s} {c:12s} n={len(vals):4d} mean={sum(vals)/len(vals):.6g}")
