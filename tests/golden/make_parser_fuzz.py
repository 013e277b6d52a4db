#!/usr/bin/env python3
"""tests/golden/parser_fuzz.json: what the REFERENCE program (oracle/_ref/modutils_ref, built from /root/reference by oracle/Makefile)
prints for the texts of tests/parser_fuzz.py -- `modutils -c 20 k w 17 -a <file>`: the "added N sequences total length L total hashes
H, new max M" line, the return code, and the lines of stderr that say what went wrong.  Run where the reference tree is."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import parser_fuzz as pf

REF = os.path.join(ROOT, "oracle", "_ref", "modutils_ref")
out = {"command": "modutils_ref -c <bits> <k> <w> 17 -a <file>", "trials": []}
with tempfile.TemporaryDirectory() as d:
    for seed in range(pf.N_TRIALS):
        kind, text = pf.make_text(seed)
        bits, k, w = pf.params(seed)
        path = os.path.join(d, "t.fq" if kind == "fastq" else "t.fa")
        open(path, "wb").write(text)
        r = subprocess.run([REF, "-c", str(bits), str(k), str(w), "17", "-a", path], capture_output=True, timeout=120)
        so, se = r.stdout.decode("latin1"), r.stderr.decode("latin1")
        added = [l for l in so.splitlines() if l.startswith("added ")]
        err = [l.replace(path, "FILE") for l in se.splitlines() if l.startswith("FATAL ERROR") or l.startswith("incomplete sequence record")]
        out["trials"].append({"seed": seed, "kind": kind, "bytes": len(text), "sha1": pf.digest(text), "k": k, "w": w, "bits": bits,
                              "rc": r.returncode, "added": added[-1] if added else None, "err": err})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "parser_fuzz.json"), "w"), indent=0)
t = out["trials"]
print("%d trials: %d fasta, %d fastq; %d with an 'added' line, %d with a FATAL ERROR, %d incomplete records" % (
    len(t), sum(x["kind"] == "fasta" for x in t), sum(x["kind"] == "fastq" for x in t), sum(x["added"] is not None for x in t),
    sum(any(e.startswith("FATAL") for e in x["err"]) for x in t), sum(any(e.startswith("incomplete") for e in x["err"]) for x in t)))
