#!/usr/bin/env python3
"""Per-(kernel, grid) durations from rocprofv3 --kernel-trace CSVs: a Python launch loop is host-bound below ~15 us per
call, so short kernels are timed from their own start / end timestamps.
    python tools/kernel_trace_summary.py [--last K] <rocprof output dir> [...]
--last K: only the last K launches of each (kernel, grid) in time order, with their mean -- bench.py's timed steps come after
its warm-up and its placement probe (the same kernel on other buffers), which the profiler records too."""
import csv
import glob
import os
import sys


def main():
    args = sys.argv[1:]
    last = 0
    if args and args[0] == "--last":
        last, args = int(args[1]), args[2:]
    for d in args:
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            groups = {}
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if "fk::" not in name:
                    continue
                key = (name.split("(")[0][-70:], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
                groups.setdefault(key, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            for key, tv in groups.items():
                tv = sorted(tv)
                if last:
                    tv = tv[-last:]
                    print(os.path.basename(d.rstrip("/")), key[0], "grid", key[1], "last %d launches: mean=%.2f us" % (len(tv), sum(x[1] for x in tv) / len(tv)))
                v = sorted(x[1] for x in tv)
                print(os.path.basename(d.rstrip("/")), key[0], "grid", key[1], "wg", key[2], "n=%d median=%.2f us min=%.2f us" % (len(v), v[len(v) // 2], v[0]))


if __name__ == "__main__":
    main()
