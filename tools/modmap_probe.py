#!/usr/bin/env python3
"""dev probe: modmap-style end to end from host byte buffers (reference build + queryProcess lines to /dev/null)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
L = mg.lib()
rng = np.random.default_rng(7)
G = 100_000_000
genome = rng.integers(0, 4, G).astype(np.uint8)
nseq = 10
goffs = (np.arange(nseq + 1) * (G // nseq)).astype(np.int64); goffs[-1] = G
sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 26)
ref = L.mgReferenceCreate(ms, 1 << 26)
names = (C.c_char_p * nseq)(*[("chr%d" % i).encode() for i in range(nseq)])
t0 = time.time()
with mg.CFile("/dev/null", "w") as f:
    L.mgReferenceRead(ref, genome.ctypes.data, goffs.ctypes.data, nseq, names, True, f)
print("reference build 100 Mbp: %.2f s" % (time.time() - t0))
total = 1_000_000_000
lens = np.clip(rng.lognormal(np.log(20000) - 0.36, 0.6, int(total / 16000)).astype(np.int64), 500, 200000)
lens = lens[np.cumsum(lens) <= total]
starts = rng.integers(0, G - 200000, len(lens))
reads = np.concatenate([genome[s:s + l] for s, l in zip(starts, lens)])
err = rng.random(len(reads)) < 0.05
reads[err] = (reads[err] + rng.integers(1, 4, int(err.sum()))) & 3
offs = np.zeros(len(lens) + 1, np.int64); offs[1:] = np.cumsum(lens)
qn = (C.c_char_p * len(lens))(*[("r%d" % i).encode() for i in range(len(lens))])
for rep in range(2):
    t0 = time.time()
    with mg.CFile("/dev/null", "w") as f:
        L.mgQueryProcess(ref, reads.ctypes.data, offs.ctypes.data, len(lens), qn, f)
    dt = time.time() - t0
    print("query %d reads, %.2f Gbp: %.2f s  %.2f Gbp/s" % (len(lens), offs[-1] / 1e9, dt, offs[-1] / dt / 1e9))
