#!/usr/bin/env python3
"""HBM bytes per launch of EVERY fk:: kernel of one command from its two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs,
never combined with a trace domain), reduced as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (KiB; FETCH_SIZE doubled).
    python tools/pmc_configs.py <fetch dir> <write dir> [configs_all.jsonl]
With the jsonl of the same command (tools/bench_configs.py rows: kernel label, units, bytes per unit) the rows are matched by launch
order is NOT attempted: the table is per kernel NAME; compare with algorithmic bytes by hand or through --expect name=bytes."""
import collections
import csv
import glob
import json
import os
import re
import sys


def collect(d, name):
    v = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if r.get("Counter_Name") == name and "fk::" in k:
                v[re.sub(r"\s+", " ", k)[:110]].append(float(r["Counter_Value"]))
    return v


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    expect = dict(a.split("=", 1)[1].rsplit("=", 1) for a in sys.argv[3:] if a.startswith("--expect="))
    for k in sorted(set(fetch) & set(write)):
        f, w = fetch[k], write[k]
        # (launches of different sizes under one name -- warm-up, parity checks -- would blur a mean: the MODE of the write size picks the bench shape)
        wm = collections.Counter(round(x) for x in w).most_common(1)[0][0]
        idx = [i for i, x in enumerate(w) if round(x) == wm]
        fm = [f[i] for i in idx if i < len(f)]
        if not fm:
            continue
        fk = sum(fm) / len(fm)
        row = {"kernel": k, "launches": len(idx), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": float(wm),
               "hbm_bytes_per_launch": int(round((2.0 * fk + wm) * 1024.0))}
        for name, b in expect.items():
            if name in k:
                row["algorithmic_bytes"] = float(b)
                row["ratio"] = row["hbm_bytes_per_launch"] / float(b)
        print(json.dumps(row))


if __name__ == "__main__":
    main()
