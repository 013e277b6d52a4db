#!/usr/bin/env python3
"""dev probe: pathological inputs (one k-mer repeated tens of millions of times; every start a modimizer)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
from oracle import pyoracle as orc
L = mg.lib()
rng = np.random.default_rng(0)
for k, w, polyA, rnd in ((21, 1, 30_000_000, 5_000_000), (21, 64, 50_000_000, 20_000_000), (15, 1, 0, 40_000_000)):
    reads = []
    if polyA: reads.append(np.zeros(polyA, np.uint8))
    reads.append(rng.integers(0, 4, rnd).astype(np.uint8))
    if polyA: reads.append(np.full(polyA // 3, 3, np.uint8))          # poly-T: the same canonical k-mer
    bases = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
    sh = mg.seqhashCreate(k, w, 17); ms = mg.modsetCreate(sh, 28)
    t0 = time.time()
    n = mg.add_sequence_batch(ms, bases, offs)
    L.modsetSyncToHost(ms, 0)
    dt = time.time() - t0
    m = ms.contents.max
    depth = np.ctypeslib.as_array(ms.contents.depth, (m + 1,))
    print("k=%d w=%d polyA=%d rnd=%d: %d hashes, %d entries, max depth %d, %.2f s" % (k, w, polyA, rnd, n, m, depth.max(), dt))
    # oracle on a bounded part (first 2 Mbp of every read) for the entry order
    h = orc.Hasher(k, w, 17); oms = orc.Modset(h, 28)
    small = [r[:2_000_000] for r in reads]
    sb = np.concatenate(small); so = np.zeros(len(small) + 1, np.int64); so[1:] = np.cumsum([len(r) for r in small])
    sh2 = mg.seqhashCreate(k, w, 17); ms2 = mg.modsetCreate(sh2, 28)
    n2 = mg.add_sequence_batch(ms2, sb, so); L.modsetSyncToHost(ms2, 0)
    tot = sum(oms.add_sequence(r) for r in small)
    m2 = ms2.contents.max
    ok = (tot == n2 and oms.max == m2 and np.array_equal(np.ctypeslib.as_array(ms2.contents.value, (m2 + 1,))[1:], oms.values()[1:])
          and np.array_equal(np.ctypeslib.as_array(ms2.contents.depth, (m2 + 1,))[1:], oms.depths()[1:]))
    print("   bounded check vs oracle:", "ok" if ok else "MISMATCH", tot, n2, oms.max, m2)
    L.modsetDestroy(ms); L.modsetDestroy(ms2); oms.close()
