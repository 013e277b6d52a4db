#!/usr/bin/env python3
"""dev: a longer randomised run of tests/test_gpu_text.py's comparisons (device parser against host parser): random record counts,
kinds, line ends, window and batch sizes.  usage: fuzz_text.py [seed] [trials]
   fuzz_text.py --vs-ref [first seed] [trials]: texts of tests/parser_fuzz.py beyond the committed fixture, mgAddSequenceFile (device parser,
   then host parser) against the REFERENCE program run on the spot (oracle/_ref/modutils_ref: where it was built, or shipped with the snapshot)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_text import crafted_fasta, crafted_fastq, device_records, last_line_is_header
from tests.test_seqio import parse_file
import modimizer_amd as mg

if len(sys.argv) > 1 and sys.argv[1] == "--vs-ref":
    import subprocess
    from tests import parser_fuzz as pf
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "modutils_ref")
    L = mg.lib(); bad = fatal = 0
    with tempfile.TemporaryDirectory() as d:
        for seed in range(first, first + trials):
            kind, text = pf.make_text(seed); bits, k, w = pf.params(seed)
            path = os.path.join(d, "t.fq" if kind == "fastq" else "t.fa"); open(path, "wb").write(text)
            r = subprocess.run([REF, "-c", str(bits), str(k), str(w), "17", "-a", path], capture_output=True, timeout=120)
            want = [l for l in r.stdout.decode("latin1").splitlines() if l.startswith("added ")]
            if r.returncode != 0 or not want:
                fatal += 1; continue                         # (the fatal cases are the fixture's business: they end the process)
            for host in ("0", "1"):
                with mg.knobs(TEXT_HOST=host):
                    sh = mg.seqhashCreate(k, w, 17); ms = mg.modsetCreate(sh, bits)
                    out = os.path.join(d, "o.txt")
                    with mg.CFile(out, "w") as f:
                        rc = L.mgAddSequenceFile(ms, path.encode(), f)
                    got = open(out).read().strip()
                    L.modsetDestroy(ms); L.mgSeqhashDestroy(sh)
                if rc or got != want[-1]:
                    bad += 1
                    print("MISMATCH seed %d (%s, %s parser): reference %r, library %r" % (seed, kind, "host" if host == "1" else "device", want[-1], got))
    print("fuzz_text --vs-ref seeds %d..%d: %d texts the reference accepts (both parsers each), %d it refuses, %d mismatches" % (first, first + trials - 1, trials - fatal, fatal, bad))
    sys.exit(1 if bad else 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(seed)
bad = 0
with tempfile.TemporaryDirectory() as d:
    for t in range(trials):
        fq = rng.random() < 0.5
        n_rec = int(rng.choice([1, 2, 7, 100, 1000, 5000]))
        if fq:
            crlf = bool(rng.random() < 0.3)
            text = crafted_fastq(rng, n_rec, crlf, int(rng.choice([5, 40, 300, 3000])))
        else:
            text = crafted_fasta(rng, n_rec, str(rng.choice(["mixed", "tiny", "long"])))
            if last_line_is_header(text): text += b"ACGT\n"       # (a file ending in a header line goes to the host parser: tests/test_gpu_text.py)
        path = os.path.join(d, "t.fq" if fq else "t.fa")
        open(path, "wb").write(text)
        _, want = parse_file(path, 1 << 40, 4)
        env = {}
        w = int(rng.choice([0, 4, 8, 12, 64, 1024])); b = int(rng.choice([0, 1, 500, 3000, 100000]))
        if w: env["MODGPU_TEXT_WINDOW_KB"] = str(w)
        if b: env["MODGPU_FILE_BATCH_BASES"] = str(b)
        os.environ.update(env); mg.lib().mgReloadKnobs()           # (the library reads its knobs once)
        try:
            rc, got = device_records(path)
        finally:
            for k in env: del os.environ[k]
            mg.lib().mgReloadKnobs()
        ok = rc == 0 and len(got) == len(want) and all(np.array_equal(a & 3, c & 3) if fq else np.array_equal(a, c) for a, c in zip(got, want))
        if not ok:
            bad += 1
            keep = "/tmp/fuzz_text_fail_%d_%d.%s" % (seed, t, "fq" if fq else "fa")
            open(keep, "wb").write(text)
            print("MISMATCH trial %d: %s n_rec %d window %d batch %d rc %d (%s) -> %s" % (t, "fastq" if fq else "fasta", n_rec, w, b, rc, mg.lib().mgLastError(), keep))
print("fuzz_text seed %d: %d trials, %d mismatches" % (seed, trials, bad))
sys.exit(1 if bad else 0)
