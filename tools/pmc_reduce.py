#!/usr/bin/env python3
"""Reduce the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, never combined with a trace domain) of one
command to HBM bytes per launch of one kernel, as MI355X_MICROARCH.md prescribes for gfx950: counters are KiB, FETCH_SIZE
reports half of a wide coalesced stream and is doubled.

    python tools/pmc_reduce.py <fetch dir> <write dir> <kernel substring> [--update-traffic aos|soa --round r03]
prints the reduction as JSON; --update-traffic writes it into profiles/pmc_traffic.json (what bench.py reports as
roofline.traffic, with its source)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def mean_counter(d, name, kernel):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == name and kernel in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    fetch_dir, write_dir, kernel = a[0], a[1], a[2]
    fk, nf = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
    wk, nw = mean_counter(write_dir, "WRITE_SIZE", kernel)
    if fk is None or wk is None:
        raise SystemExit(f"no counters for {kernel!r} ({nf} fetch rows, {nw} write rows)")
    out = {"kernel": kernel, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "launches": [nf, nw],
           "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024.0))}
    print(json.dumps(out))
    if "--update-traffic" in sys.argv:
        lay = sys.argv[sys.argv.index("--update-traffic") + 1]
        rnd = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r03"
        p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        cur = json.load(open(p))
        cur[lay] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "round": rnd,
                    "source": f"profiles/{rnd}/kf_c2_{lay}_pmc_fetch.csv + kf_c2_{lay}_pmc_write.csv (mean of {nf} / {nw} launches of "
                              f"`python bench.py --steps 20 --warmup 5`, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_reduce.py)"}
        json.dump(cur, open(p, "w"), indent=2)


if __name__ == "__main__":
    main()
