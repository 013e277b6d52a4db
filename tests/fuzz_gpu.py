#!/usr/bin/env python3
"""dev fuzz: random (k, d, seed) and ragged read batches, GPU scan / modset / minimizers against the oracle"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
from oracle import pyoracle as orc
L = mg.lib()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(seed0)
t0 = time.time(); bad = 0
for t in range(trials):
    k = int(rng.integers(1, 32)); w = int(rng.choice([1, 2, 3, 4, 8, 16, 31, 32, 64, 100, 128, 255, 1000]))
    if w > 64 and k < 4: w = 7
    seed = int(rng.integers(1, 1000))
    n_reads = int(rng.integers(1, 60))
    kinds = rng.integers(0, 6, n_reads)
    reads = []
    for kind in kinds:
        if kind == 0: L_ = int(rng.integers(0, 3 * k + 2))
        elif kind == 1: L_ = int(rng.integers(4090, 4102))              # around a tile
        elif kind == 2: L_ = int(rng.integers(100, 3000))
        elif kind == 3: L_ = int(rng.integers(8000, 40000))
        elif kind == 4: L_ = int(rng.choice([4096 - k, 4096, 4096 + k - 1, 8192, 4096 * 3 + 5]))
        else: L_ = int(rng.integers(1, 300))
        r = rng.integers(0, 4, L_).astype(np.uint8)
        if rng.random() < 0.15 and L_ > 50: r[int(L_ * 0.3):int(L_ * 0.7)] = rng.integers(0, 4)      # homopolymer stretch
        reads.append(r)
    bases = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
    offs = np.zeros(len(reads) + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
    h = orc.Hasher(k, w, seed); sh = mg.seqhashCreate(k, w, seed)
    want = [h.scan(r) for r in reads]
    km, pos, isf, st = mg.scan_batch(sh, bases, offs)
    ok = st[-1] == sum(len(x[0]) for x in want)
    for r, (a, b, c) in enumerate(want):
        s, e = st[r], st[r + 1]
        ok = ok and np.array_equal(km[s:e], a) and np.array_equal(pos[s:e], b) and np.array_equal(isf[s:e], c)
    if t % 3 == 0 and len(bases) / w < 200_000:                      # modset build (stays under the 2^18 entries of 20 table bits)
        bits = 20
        oms = orc.Modset(h, bits); tot = sum(oms.add_sequence(r) for r in reads)
        ms = mg.modsetCreate(sh, bits)
        # the reads in one to three batches, with a lookup of present and absent k-mers after the first
        cuts = sorted(set([0, len(reads)] + [int(c) for c in rng.integers(0, len(reads) + 1, int(rng.integers(0, 3)))]))
        n = 0
        for bi in range(len(cuts) - 1):
            a, b = cuts[bi], cuts[bi + 1]
            if a == b: continue
            n += mg.add_sequence_batch(ms, bases[offs[a]:offs[b]], offs[a:b + 1] - offs[a])
            if bi == 0 and ms.contents.max > 0 and k <= 31:
                import ctypes as C
                L.modsetSyncToHost(ms, 0)
                vv = np.ctypeslib.as_array(ms.contents.value, (ms.contents.max + 1,))[1:].copy()
                probe = np.concatenate([vv[::3], (vv[::5] ^ np.uint64(5)) & np.uint64((1 << (2 * k)) - 1)])
                d_p = mg.DeviceBuffer.from_numpy(probe); d_o = mg.DeviceBuffer(len(probe) * 4)
                mg.check(L.modsetFindBatchDevice(ms, d_p.ptr, len(probe), d_o.ptr, None))
                got = d_o.to_numpy(np.uint32, len(probe))
                index_of = {int(x): i + 1 for i, x in enumerate(vv)}
                ok = ok and all(int(got[i]) == index_of.get(int(probe[i]), 0) for i in range(len(probe)))
                d_p.free(); d_o.free()
        L.modsetSyncToHost(ms, 1)
        v, d, _ = mg.modset_arrays(ms)
        ok = ok and n == tot and ms.contents.max == oms.max and np.array_equal(v[1:], oms.values()[1:]) and np.array_equal(d[1:], oms.depths()[1:])
        ok = ok and np.array_equal(np.ctypeslib.as_array(ms.contents.index, (1 << bits,)), oms.index_table())
        L.modsetDestroy(ms); oms.close()
    if t % 4 == 1 and w <= 300:                                      # minimizers
        wantm = [h.minimizers(r) for r in reads]
        hv, p2, f2, s2 = mg.minimizer_batch(sh, bases, offs)
        for r, (a, b, c) in enumerate(wantm):
            s, e = s2[r], s2[r + 1]
            ok = ok and np.array_equal(hv[s:e], a) and np.array_equal(p2[s:e], b) and np.array_equal(f2[s:e], c)
    if not ok:
        bad += 1
        print("MISMATCH trial", t, "k", k, "w", w, "seed", seed, "reads", [len(r) for r in reads][:12])
print("fuzz seed %d: %d trials, %d mismatches, %.1f s" % (seed0, trials, bad, time.time() - t0))
