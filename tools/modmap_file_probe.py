#!/usr/bin/env python3
"""dev probe: mgReferenceFastaRead + mgQueryFile on a 150-base FASTQ file (and a 10 kb FASTA file) through the device parser and the
host parser, with the parser's own phase times (MODGPU_TEXT_TIMING=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); mg.check(L.mgSetDevice(0))
shm = "/dev/shm"
G = 20_000_000
g = synth.iid_bases(G, 5)
letters = np.frombuffer(b"ACGT", np.uint8)
rpath = os.path.join(shm, "probe_ref.fa")
with open(rpath, "wb") as f:
    f.write(b">ref1\n"); f.write(np.concatenate([letters[g].reshape(-1, 80), np.full((G // 80, 1), 10, np.uint8)], axis=1).tobytes())
nq = int(os.environ.get("PROBE_READS", "4000000"))
rng = np.random.default_rng(1)
st = rng.integers(0, G - 150, nq)
seqs = letters[g[(st[:, None] + np.arange(150)[None, :])]]
qpath = os.path.join(shm, "probe_q.fq")
hdr = np.array([list(b"@read%09d\n" % i) for i in range(0, 1)], np.uint8)
ids = np.char.add(np.char.add("@r", np.arange(nq).astype(str)), "\n")
with open(qpath, "wb") as f:
    blk = []
    for i in range(0, nq, 200000):
        part = b"".join(ids[j].encode() + seqs[j].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for j in range(i, min(nq, i + 200000)))
        f.write(part)
for host in ("0", "1"):
    with mg.knobs(TEXT_HOST=host, TEXT_TIMING=1, SEED_TIMING=1):
        sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 24)
        ref = L.mgReferenceCreate(ms, 1 << 26)
        with mg.CFile(os.devnull, "w") as fo:
            assert L.mgReferenceFastaRead(ref, rpath.encode(), True, fo) == 0
        for it in range(3):
            out = os.path.join(shm, "probe_out.txt")
            if os.path.exists(out):
                os.remove(out)                      # (truncating 235 MB of tmpfs inside the timed call costs 15 ms)
            t0 = time.perf_counter()
            with mg.CFile(out, "w") as fo:
                assert L.mgQueryFile(ref, qpath.encode(), fo) == 0
            dt = time.perf_counter() - t0
            print("host parser" if host == "1" else "device parser", "run", it, "%.3f s" % dt, "%.2f Gbp/s" % (nq * 150 / dt / 1e9), "%.1f M lines/s" % (nq / dt / 1e6), os.path.getsize(out), flush=True)
        # the call's fixed cost: a file of 1000 reads
        small = os.path.join(shm, "probe_small.fq")
        with open(qpath, "rb") as f:
            open(small, "wb").write(b"".join(f.readline() for _ in range(4000)))
        for it in range(3):
            t0 = time.perf_counter()
            with mg.CFile(out, "w") as fo:
                assert L.mgQueryFile(ref, small.encode(), fo) == 0
            print("   1000 reads: %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        os.remove(small)
        L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
os.remove(rpath); os.remove(qpath)
