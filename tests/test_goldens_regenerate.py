"""The committed fixtures ARE what the live reference produces: tools/verify_goldens.py re-runs the golden generators on a copy of
tests/ and compares array by array, bit for bit.  Where the reference checkout exists (the build container) every generator but
the 30-second one runs in the default CPU suite (11 generators, 11 fixtures, ~12 s); FK_VERIFY_ALL_GOLDENS=1 adds
make_goldens.py (the ten largest fixtures; `python tools/verify_goldens.py` by hand does the same:
profiles/r06/goldens_verified.json).  Elsewhere: skipped."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = os.environ.get("FILTERPY_REFERENCE", "/root/reference")


def test_committed_fixtures_regenerate_bit_identically():
    if not os.path.isdir(os.path.join(REF, "filterpy")):
        pytest.skip("no reference checkout here")
    gens = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "make_*.py")))
    if not os.environ.get("FK_VERIFY_ALL_GOLDENS"):
        gens = [g for g in gens if g != "make_goldens.py"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_goldens.py"), "--reference", REF, "--only", ",".join(gens)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    assert "ALL bit-identical" in r.stdout and r.stdout.count("bit-identical") >= len(gens)
