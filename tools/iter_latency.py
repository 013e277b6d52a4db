#!/usr/bin/env python3
"""dev probe: per-call cost of the iterator facade (modRCiterator + drain + destroy) by read length"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
L = mg.lib(); mg.check(L.mgSetDevice(0))
sh = mg.seqhashCreate(21, 64, 17)
rng = np.random.default_rng(1)
for n in (150, 1000, 10000, 100000, 262144, 1000000):
    b = rng.integers(0, 4, n).astype(np.uint8)
    reps = 2000 if n <= 10000 else 200
    for _ in range(20):
        it = L.modRCiterator(sh, b.ctypes.data, n); L.mgSeqhashRCiteratorDestroy(it)
    t = time.perf_counter()
    for _ in range(reps):
        it = L.modRCiterator(sh, b.ctypes.data, n); L.mgSeqhashRCiteratorDestroy(it)
    dt = (time.perf_counter() - t) / reps
    print("read length %8d: %8.2f us per modRCiterator call = %8.1f Mbp/s" % (n, dt * 1e6, n / dt / 1e6))
