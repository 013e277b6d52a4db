"""A short run of tools/soak.py inside the GPU suite: the randomized differential soak (scan / build / query / minimizer / modmap / read
ingest / pipelined query against the oracle, with drawn k, w, seed, table bits and path knobs) from a fixed first seed, so that the trials
are the same on every box; the long runs are recorded in profiles/r05_soak.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_soak_short_fixed_seed():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "25", "2024"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0 and "SOAK_OK" in r.stdout, tail
    n = int(r.stdout.rsplit("SOAK_OK", 1)[1].split()[0])
    assert n >= 70, "fewer trials than ten of each kind: %d" % n
