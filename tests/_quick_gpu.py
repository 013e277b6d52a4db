"""quick first-contact GPU check (not a pytest file)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
from modimizer_amd import synth
from oracle import pyoracle as po

L = mg.lib()
print("devices", L.mgDeviceCount(), L.mgVersion())
for (k, d) in [(21, 64), (31, 4), (19, 31), (1, 1), (16, 32)]:
    sh = mg.seqhashCreate(k, d, 17)
    oh = po.Hasher(k, d, 17)
    rng = np.random.default_rng(k * 1000 + d)
    lens = [0, 1, k - 1, k, k + 1, 63, 64, 65, 100, 1000, 16384, 16383 + k, 50000, 3]
    offs = np.zeros(len(lens) + 1, np.int64); offs[1:] = np.cumsum(lens)
    bases = rng.integers(0, 4, int(offs[-1])).astype(np.uint8)
    t = time.time()
    km, pos, isf, st = mg.scan_batch(sh, bases, offs)
    dt = time.time() - t
    ok = True
    for r, Ln in enumerate(lens):
        a, b, c = oh.scan(bases[offs[r]:offs[r + 1]])
        s, e = st[r], st[r + 1]
        if not (np.array_equal(a, km[s:e]) and np.array_equal(b, pos[s:e]) and np.array_equal(c, isf[s:e])):
            ok = False; print("MISMATCH", k, d, r, Ln, len(a), e - s)
    print("scan", k, d, "n", len(km), "ok" if ok else "FAIL", "%.3fs" % dt)
    # iterator facade
    a, b, c = mg.iterate(sh, bases[offs[9]:offs[10]])
    x, y, z = oh.scan(bases[offs[9]:offs[10]])
    print("  iterator", np.array_equal(a, x) and np.array_equal(b, y) and np.array_equal(c, z))

# modset build
k, d = 21, 64
sh = mg.seqhashCreate(k, d, 17); oh = po.Hasher(k, d, 17)
genome = synth.iid_bases(200000, 5)
starts, offsets, strands = synth.ont_read_plan(3_000_000, len(genome), 7, n50=5000, lo=100, hi=20000)
bases = synth.reads_from_genome(genome, starts, offsets, strands, 0.02, 11)
ms = mg.modsetCreate(sh, 24); oms = po.Modset(oh, 24)
n = mg.add_sequence_batch(ms, bases, offsets.astype(np.int64))
tot = 0
for r in range(len(starts)):
    tot += oms.add_sequence(bases[int(offsets[r]):int(offsets[r + 1])])
mg.check(L.modsetSyncToHost(ms, 1))
v, dp, info = mg.modset_arrays(ms)
print("modset hashes", n, tot, "max", ms.contents.max, oms.max)
print("  values", np.array_equal(v[1:], oms.values()[1:]), "depths", np.array_equal(dp[1:], oms.depths()[1:]))
idx = np.ctypeslib.as_array(ms.contents.index, (1 << 24,))
print("  index table", np.array_equal(idx, oms.index_table()))
