#!/usr/bin/env python3
"""Mean of every counter of rocprofv3 --pmc passes, per kernel:
    python tools/pmc_summary.py <dir> [<dir> ...] [--kernel substr | --all]
One JSON line per kernel with the mean of each counter over its launches -- small enough to commit next to the bench line of
the same box.  Default: the headline kernel (kf_fast); --all: every kernel of the library (fk::)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    dirs = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--kernel"]
    want = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else ("fk::" if "--all" in sys.argv else "kf_fast")
    acc = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if want in k:
                    acc[(k.split("(")[0][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    by_kernel = defaultdict(dict)
    for (k, c), v in acc.items():
        by_kernel[k][c] = {"mean": sum(v) / len(v), "launches": len(v)}
    for k, cs in by_kernel.items():
        print(json.dumps({"kernel": k, "counters": {c: round(x["mean"], 1) for c, x in sorted(cs.items())},
                          "launches": max(x["launches"] for x in cs.values())}))
    if not by_kernel:
        print(json.dumps({"error": "no counter rows for " + want, "dirs": dirs}))


if __name__ == "__main__":
    main()
