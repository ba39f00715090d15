#!/usr/bin/env python3
"""Per-(kernel, grid) durations from rocprofv3 --kernel-trace CSVs: a Python launch loop is host-bound below ~15 us per
call, so short kernels are timed from their own start / end timestamps.
    python tools/kernel_trace_summary.py <rocprof output dir> [...]"""
import csv
import glob
import os
import sys


def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            groups = {}
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if "fk::" not in name:
                    continue
                key = (name.split("(")[0][-70:], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
                groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            for key, v in groups.items():
                v = sorted(v)
                print(os.path.basename(d.rstrip("/")), key[0], "grid", key[1], "wg", key[2], "n=%d median=%.2f us min=%.2f us" % (len(v), v[len(v) // 2], v[0]))


if __name__ == "__main__":
    main()
