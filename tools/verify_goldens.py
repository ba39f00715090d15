#!/usr/bin/env python3
"""Re-run every committed golden generator against the LIVE reference and compare with the committed fixtures, array by array,
bit for bit ("pinned").  One command for what a reviewer otherwise does by hand:

    python tools/verify_goldens.py [--reference /root/reference] [--only make_goldens.py,...] [--json out.json]

The generators (tests/golden/make_*.py) write beside themselves, so they run on a COPY of tests/ (scripts + the helper modules they
import) in a temporary directory; the committed files are never touched.  Needs the reference checkout -- the build container only;
exit code 0 = every array of every fixture identical, 1 = a difference (listed), 2 = the reference is not there.
Test infrastructure: nothing in filterpy_amd/ imports this."""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def same_array(a, b):
    """bit-identical: same dtype, shape and bytes (NaNs compare by their bits; object arrays by value)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype == object:
        return bool(np.all(a == b))
    return a.tobytes() == b.tobytes()


def compare_npz(new, old):
    with np.load(new, allow_pickle=True) as A, np.load(old, allow_pickle=True) as B:
        ka, kb = set(A.files), set(B.files)
        bad = sorted(ka ^ kb)
        for k in sorted(ka & kb):
            if not same_array(A[k], B[k]):
                bad.append(k)
        return len(ka | kb), bad


def compare_json(new, old):
    with open(new) as fa, open(old) as fb:
        A, B = json.load(fa), json.load(fb)
    if A == B:
        return (len(A) if hasattr(A, "__len__") else 1), []
    if isinstance(A, dict) and isinstance(B, dict):
        return len(set(A) | set(B)), sorted(k for k in set(A) | set(B) if A.get(k) != B.get(k))
    return 1, ["<document>"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("FILTERPY_REFERENCE", "/root/reference"))
    ap.add_argument("--only", default="", help="comma-separated generator file names (default: all)")
    ap.add_argument("--json", default="", help="write the per-fixture summary here")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.reference, "filterpy")):
        print(f"verify_goldens: no reference at {args.reference} (this check runs in the build container only)")
        return 2
    gens = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "make_*.py")))
    if args.only:
        gens = [g for g in gens if g in set(args.only.split(","))]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg", FILTERPY_REFERENCE=args.reference,
               OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1",
               PYTHONPATH=args.reference + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else ""))
    rows, failed = [], False
    with tempfile.TemporaryDirectory(prefix="fk_goldens_") as tmp:
        tdir = os.path.join(tmp, "tests")
        os.makedirs(os.path.join(tdir, "golden"))
        for p in glob.glob(os.path.join(ROOT, "tests", "*.py")):          # helper modules the generators import (ukf_hook_model, ...)
            shutil.copy(p, tdir)
        for p in glob.glob(os.path.join(GOLD, "*")):                      # the generators, and the committed fixtures some of them
            if os.path.isfile(p):                                         # read as INPUT (make_conditioning.py: ukf_merwe.npz)
                shutil.copy(p, os.path.join(tdir, "golden"))
        for g in gens:
            before = {f: os.path.getmtime(os.path.join(tdir, "golden", f)) for f in os.listdir(os.path.join(tdir, "golden"))}
            t0 = time.time()
            r = subprocess.run([sys.executable, os.path.join(tdir, "golden", g)], cwd=tmp, env=env, capture_output=True, text=True)
            dt = time.time() - t0
            if r.returncode != 0:
                rows.append({"generator": g, "error": (r.stderr or r.stdout)[-400:], "seconds": round(dt, 1)})
                failed = True
                print(f"{g}: FAILED to run\n{(r.stderr or r.stdout)[-400:]}")
                continue
            made = sorted(f for f in os.listdir(os.path.join(tdir, "golden"))
                          if f.endswith((".npz", ".json")) and before.get(f) != os.path.getmtime(os.path.join(tdir, "golden", f)))
            if not made:
                rows.append({"generator": g, "error": "wrote no fixture", "seconds": round(dt, 1)})
                failed = True
                print(f"{g}: wrote no fixture")
            for f in made:
                old = os.path.join(GOLD, f)
                if not os.path.exists(old):
                    rows.append({"generator": g, "fixture": f, "error": "not committed"})
                    failed = True
                    print(f"{g}: {f} is not a committed fixture")
                    continue
                n, bad = (compare_npz if f.endswith(".npz") else compare_json)(os.path.join(tdir, "golden", f), old)
                rows.append({"generator": g, "fixture": f, "entries": n, "different": bad, "seconds": round(dt, 1)})
                failed = failed or bool(bad)
                print(f"{g}: {f}: {n} entries, " + ("bit-identical" if not bad else f"{len(bad)} DIFFERENT: {bad[:8]}"), flush=True)
    covered = {r.get("fixture") for r in rows}
    orphans = sorted(f for f in os.listdir(GOLD) if f.endswith((".npz", ".json")) and f not in covered) if not args.only else []
    for f in orphans:
        print(f"(no generator rewrote {f})")
    total = sum(r.get("entries", 0) for r in rows)
    print(f"verify_goldens: {len([r for r in rows if 'fixture' in r])} fixtures, {total} entries, "
          + ("ALL bit-identical to the committed files" if not failed else "DIFFERENCES (see above)"))
    if args.json:
        import numpy
        import scipy
        with open(args.json, "w") as fh:
            json.dump({"reference": args.reference, "numpy": numpy.__version__, "scipy": scipy.__version__, "ok": not failed,
                       "fixtures": rows, "not_regenerated": orphans}, fh, indent=1)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
