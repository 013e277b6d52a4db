#!/usr/bin/env python3
"""dev: randomized differential soak -- the library's GPU paths against the oracle with parameters the fixed test matrix does not hold.
Every trial draws (k, w, seed, table bits, knobs) and a read set with the awkward reads mixed in (shorter than k, exactly k, empty,
homopolymers, tandem repeats, reverse-complement palindromes, reads that end on tile edges), then checks, bit for bit:
  scan      seqhashScanBatchDevice: k-mers, positions, strands, per-read counts              (seqhash.c:154-196)
  build     mgAddSequenceBatch in 1-3 batches: value[] / depth[] / index[] and the hash totals  (modutils.c:19-31, modset.c:45-62)
  query     mgQueryReadsDevice on mutated reads, one of the lookup paths: Seed{index, pos}       (modmap.c:197-206)
  minimizer seqhashMinimizerBatch                                                              (seqhash.c:83-152)
  modmap    mgReferenceRead + mgQueryProcess against the oracle's queryProcess, byte for byte   (modmap.c:74-134,188-281)
  readset   mgReadsetRead / mgReadsetFileRead against the oracle's Readset: hit lists, distances, inverse lists, depth[], RS lines (modasm.c:151-287)
  pipelined mgQueryReadsDeviceAsync / Wait with two batches in flight == mgQueryReadsDevice batch by batch
usage: soak.py [seconds (default 300)] [first seed (default: from the clock)]
Prints one line per trial and, at the end, 'SOAK_OK <trials>' -- or the failing trial's seed (re-run with it as the second argument)."""
import ctypes as C, os, sys, tempfile, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import modimizer_amd as mg
from oracle import pyoracle as po
import util
import test_gpu_scan as tscan
import test_gpu_modset as tmod

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0xffffff
L = mg.lib(); mg.check(L.mgSetDevice(0))
TILE = tscan.TILE
LARGE = os.environ.get("SOAK_LARGE") == "1"        # build trials only, at sizes where every stage of the bucketed build has many workgroups


def draw_params(rng):
    k = int(rng.choice([1, 2, 3, 5, 8, 11, 13, 15, 16, 17, 19, 20, 21, 24, 27, 29, 31, int(rng.integers(1, 32))]))
    kind = rng.integers(0, 6)
    if kind == 0:
        w = 1 << int(rng.integers(0, 11))                        # powers of two: FAST where it applies, else POW2
    elif kind == 1:
        w = int(rng.choice([3, 5, 7, 11, 31, 33, 97, 127, 255, 257, 1001, 4097, 32767, 32769, 65537]))      # odd: ODD32 below 2^15 and k <= 20, else ODD
    elif kind == 2:
        w = int(rng.choice([6, 10, 12, 24, 48, 96, 100, 1000, 1024 * 3, 40000, 2 * 32769]))                 # even, no power of two: ANY32 / ANY
    elif kind == 3:
        w = int(rng.integers(1, 200))
    elif kind == 4:
        w = int(rng.integers(1, 70000))
    else:
        w = int(rng.choice([64, 31, 4]))
    return k, w, int(rng.integers(0, 1000))


def awkward_reads(rng, k, total):
    reads = []
    fixed = [0, 1, max(k - 1, 0), k, k + 1, 63, 64, 65, TILE - 1, TILE, TILE + 1, TILE + k - 1, TILE - k + 1, 2 * TILE, 2 * TILE + k - 2]
    for n in rng.permutation(fixed)[:int(rng.integers(4, len(fixed)))]:
        reads.append(rng.integers(0, 4, int(n)).astype(np.uint8))
    reads.append(np.full(int(rng.integers(k, 3000)), int(rng.integers(0, 4)), np.uint8))                    # homopolymer
    unit = rng.integers(0, 4, int(rng.integers(1, 9))).astype(np.uint8)
    reads.append(np.tile(unit, int(rng.integers(20, 800))))                                                   # tandem repeat
    half = rng.integers(0, 4, int(rng.integers(k, 400))).astype(np.uint8)
    reads.append(np.concatenate([half, (3 - half[::-1]).astype(np.uint8)]))                                   # reverse-complement palindrome: strand ties
    genome = rng.integers(0, 4, max(2000, total // 6)).astype(np.uint8)
    left = total
    while left > 0:
        n = int(min(left, max(1, rng.lognormal(np.log(4000), 0.9))))
        s = int(rng.integers(0, max(1, len(genome) - n)))
        r = genome[s:s + n].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        if rng.random() < 0.7:
            hit = rng.random(len(r)) < 0.03
            r[hit] = (r[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
        reads.append(r); left -= n
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


def trial_scan(rng, k, w, sd):
    sh = mg.seqhashCreate(k, w, sd); oh = po.Hasher(k, w, sd)
    reads = awkward_reads(rng, k, int(rng.integers(20_000, 400_000)) if w > 2 else int(rng.integers(5_000, 60_000)))
    tscan.assert_batch_equal(sh, oh, reads)


BUILD_KNOBS = [{}, {}, {"TABLE_PATH": "bucket"}, {"TABLE_PATH": "direct"}, {"TABLE_PATH": "bucket", "FLAG_POLARITY": 0}, {"TABLE_PATH": "bucket", "FLAG_POLARITY": 1},
               {"TABLE_PATH": "bucket", "PART_PACKED": 0, "FLAG_POLARITY": 1}, {"TABLE_PATH": "bucket", "BUCKET_R": 1024, "BUCKET_T": 256},
               {"TABLE_PATH": "bucket", "HOT_SPLIT": "200,64"}, {"TABLE_PATH": "bucket", "HOT_SPLIT": "200,64", "FLAG_POLARITY": 0},
               {"TABLE_PATH": "bucket", "PART_BIG": 0}, {"TABLE_PATH": "bucket", "PART_DIGITS": 0}, {"TABLE_PATH": "bucket", "MERGE_SLOTS": 0},
               {"TABLE_PATH": "bucket", "MERGE_SLOTS": 1}, {"ADD_CHUNK": 100000}, {"SCAN_GENERIC": 1}, {"SCAN_GRID": 64},
               # round 6: the table brought to a tight load after the dedup kernel's count (forced, with the slack a bucket keeps cut to nothing at 95), or never
               {"TABLE_PATH": "bucket", "TIGHT_LOAD": 70}, {"TABLE_PATH": "bucket", "TIGHT_LOAD": 95, "FLAG_POLARITY": 1}, {"TABLE_PATH": "bucket", "TIGHT_LOAD": 0},
               {"TABLE_PATH": "bucket", "TIGHT_LOAD": 50, "BUCKET_R": 2048, "BUCKET_T": 512}, {"TABLE_LOAD": 95}, {"TABLE_PATH": "bucket", "TABLE_LOAD": 30}]


def trial_build(rng, k, w, sd):
    with mg.knobs(**BUILD_KNOBS[int(rng.integers(0, len(BUILD_KNOBS)))]):
        _trial_build(rng, k, w, sd)


def _trial_build(rng, k, w, sd):
    bits = int(rng.integers(20, 24))
    sh = mg.seqhashCreate(k, w, sd); oh = po.Hasher(k, w, sd)
    tot = int(rng.integers(50_000, 600_000)) if w > 2 else int(rng.integers(20_000, 120_000))
    if rng.random() < 0.15:                                        # now and then a batch large enough for the bucketed build by itself (>= 1.5e6 modimizers is its rule)
        tot = int(min(12_000_000, max(tot, 2_000_000 * min(w, 8))))
        bits = 24
    if LARGE:                                                      # SOAK_LARGE=1: batches of 30 - 300 Mbp (3e6 .. 1e8 modimizers), table bits to match
        tot = int(rng.integers(30_000_000, 300_000_000)) if w >= 16 else int(rng.integers(10_000_000, 60_000_000))
        bits = 28
    reads = awkward_reads(rng, k, tot)
    cuts = sorted(set([0, len(reads)] + [int(x) for x in rng.integers(0, len(reads) + 1, int(rng.integers(0, 3)))]))
    ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
    n = t = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = reads[a:b]
        if not part:
            continue
        bases, offs = util.concat_reads(part)
        n += mg.add_sequence_batch(ms, bases, offs.astype(np.int64))
        t += sum(oms.add_sequence(r) for r in part)
    assert n == t, ("hash totals", n, t)
    tmod.assert_same_modset(ms, oms, bits)
    L.modsetDestroy(ms)


def trial_query(rng, k, w, sd):
    path = str(rng.choice(["direct", "part", "2 levels"]))
    bits = int(rng.choice([20, 22, 24]))
    with mg.knobs(FIND_PATH=path):
        sh = mg.seqhashCreate(k, w, sd); oh = po.Hasher(k, w, sd)
        g = rng.integers(0, 4, int(rng.integers(30_000, 200_000))).astype(np.uint8)
        ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
        mg.add_sequence_batch(ms, g, np.array([0, len(g)], np.int64)); oms.add_sequence(g)
        reads = awkward_reads(rng, k, int(rng.integers(20_000, 200_000)) if w > 2 else 20_000)
        reads += [g[s:s + 3000].copy() for s in rng.integers(0, len(g) - 3000, 6)]
        bases, offs = util.concat_reads(reads)
        qk, qp, _, qst = util.oracle_scan_batch(oh, bases, offs)
        want = np.array([oms.find(int(x)) for x in qk], np.uint32)
        total = int(offs[-1])
        d_p = mg.DeviceBuffer.from_numpy(mg.pack_host(bases)); d_o = mg.DeviceBuffer.from_numpy(offs.astype(np.uint64))
        cap = len(qk) + 5
        d_ix = mg.DeviceBuffer(cap * 4); d_pos = mg.DeviceBuffer(cap * 4); d_rid = mg.DeviceBuffer(cap * 4)
        n = C.c_uint64()
        mg.check(L.mgQueryReadsDevice(ms, d_p.ptr, total, d_o.ptr, len(offs) - 1, d_ix.ptr, d_pos.ptr, d_rid.ptr, cap, C.byref(n), None))
        assert n.value == len(qk), (n.value, len(qk))
        assert np.array_equal(d_ix.to_numpy(np.uint32, n.value), want), "seed indices"
        assert np.array_equal(d_pos.to_numpy(np.uint32, n.value) & mg.MG_POS_MASK, qp.astype(np.uint32)), "seed positions"
        rid = d_rid.to_numpy(np.uint32, n.value)
        assert np.array_equal(np.searchsorted(rid, np.arange(len(offs))), qst), "seeds per read"
        L.modsetDestroy(ms)


def trial_minimizer(rng, k, w, sd):
    w = int(min(w, rng.choice([255, 1023]))) or 1
    sh = mg.seqhashCreate(k, w, sd); oh = po.Hasher(k, w, sd)
    reads = awkward_reads(rng, k, int(rng.integers(10_000, 120_000)))
    bases, offs = util.concat_reads(reads)
    got = mg.minimizer_batch(sh, bases, offs.astype(np.int64))
    hs, ps, fs, st = [], [], [], [0]
    for r in reads:
        h, p, f = oh.minimizers(r)
        hs.append(h); ps.append(p); fs.append(f); st.append(st[-1] + len(h))
    assert np.array_equal(got[3], np.array(st, np.int64)), "minimizers per read"
    assert np.array_equal(got[0], np.concatenate(hs).astype(got[0].dtype)), "minimizer hashes"
    assert np.array_equal(got[1], np.concatenate(ps).astype(got[1].dtype)), "minimizer positions"
    assert np.array_equal(got[2], np.concatenate(fs).astype(got[2].dtype)), "minimizer strands"


def trial_modmap(rng, k, w, sd, tmp):
    if k < 9 or w > 64:
        k, w = int(rng.choice([13, 15, 17, 19, 21, 25])), int(rng.choice([4, 8, 11, 16, 31, 64]))
    path = str(rng.choice(["direct", "part", "2 levels"]))
    with mg.knobs(FIND_PATH=path):
        try:
            tmod._modmap_randomized(L, k, w, sd, tmp)
        except AssertionError as e:                       # the helper's last two asserts are about its OWN coverage (enough M lines, one overflowing read): not a difference
            line = traceback.extract_tb(e.__traceback__)[-1].line or ""
            if "n_m_lines" not in line and "n_overflow" not in line:
                raise


def trial_readset(rng, k, w, sd, tmp):
    """modasm's ingest (mgReadsetRead, one call or a FASTA file in small batches) against the oracle's Readset: every array, depth[], the RS lines"""
    import test_readset as trs
    if w > 256:
        w = int(rng.integers(1, 64))
    h = po.Hasher(k, w, sd); oms = po.Modset(h, 20)
    g = rng.integers(0, 4, int(rng.integers(5_000, 60_000))).astype(np.uint8)
    oms.add_sequence(g)
    if oms.max == 0:
        oms.close(); return
    oms.set_copy(1, 2, 3)
    reads = awkward_reads(rng, k, int(rng.integers(5_000, 150_000)) if w > 2 else int(rng.integers(5_000, 30_000)))
    reads += [g[a:a + 2500].copy() for a in rng.integers(0, max(1, len(g) - 2500), 5)]
    ors = po.Readset(oms); ors.read(reads)
    want = ors.arrays()
    pmod = str(tmp / "m.mod"); oms.write_mod(pmod)
    with mg.CFile(pmod, "r") as f:
        ms = L.modsetRead(f)
    rs = L.mgReadsetCreate(ms)
    if rng.random() < 0.5:
        bases, offs = util.concat_reads(reads)
        assert L.mgReadsetRead(rs, bases.ctypes.data, offs.ctypes.data, len(reads)) == 0
    else:
        fa = str(tmp / "r.fa")
        with open(fa, "w") as f:
            for i, r in enumerate(reads):
                f.write(">r%d\n%s\n" % (i, "".join("ACGT"[b] for b in r)))
        with mg.knobs(FILE_BATCH_MBP=1, FILE_BATCH_BASES=int(rng.integers(3000, 200_000))):
            assert L.mgReadsetFileRead(rs, fa.encode()) == 0
    trs.same(want, trs.lib_arrays(rs))
    assert np.array_equal(np.ctypeslib.as_array(ms.contents.depth, (ms.contents.max + 1,)), oms.depths()), "depth[] after the ingest"
    assert trs.rs_lines(trs.stats_text(rs, str(tmp / "s.txt"))) == trs.rs_lines(ors.stats_text(str(tmp / "o.txt"))), "RS lines"
    L.mgReadsetDestroy(rs); L.modsetDestroy(ms)
    ors.close(); oms.close()


def trial_pipelined(rng, k, w, sd):
    """mgQueryReadsDeviceAsync / Wait over several batches (two in flight) == mgQueryReadsDevice batch by batch"""
    path = rng.choice(["direct", "2 levels", ""])
    with mg.knobs(FIND_PATH=str(path) if path else None):
        sh = mg.seqhashCreate(k, w, sd)
        g = rng.integers(0, 4, int(rng.integers(30_000, 150_000))).astype(np.uint8)
        ms = mg.modsetCreate(sh, 22)
        mg.add_sequence_batch(ms, g, np.array([0, len(g)], np.int64))
        batches = []
        for b in range(int(rng.integers(2, 6))):
            reads = awkward_reads(rng, k, int(rng.integers(1_000, 150_000)) if w > 2 else 15_000) if rng.random() < 0.85 else [np.zeros(0, np.uint8)]
            reads += [g[a:a + 2000].copy() for a in rng.integers(0, len(g) - 2000, 3)]
            bases, offs = util.concat_reads(reads)
            total = int(offs[-1]); cap = total + 16
            batches.append(dict(total=total, n=len(reads), cap=cap, p=mg.DeviceBuffer.from_numpy(mg.pack_host(bases)), o=mg.DeviceBuffer.from_numpy(offs.astype(np.uint64)),
                                ix=[mg.DeviceBuffer(cap * 4) for _ in range(2)], pos=[mg.DeviceBuffer(cap * 4) for _ in range(2)], rid=[mg.DeviceBuffer(cap * 4) for _ in range(2)]))
        n = C.c_uint64(); counts = []
        for b in batches:                                           # synchronous: results in slot 0
            mg.check(L.mgQueryReadsDevice(ms, b["p"].ptr, b["total"], b["o"].ptr, b["n"], b["ix"][0].ptr, b["pos"][0].ptr, b["rid"][0].ptr, b["cap"], C.byref(n), None))
            counts.append(n.value)
        tickets = []
        def wait_one():
            i, t = tickets.pop(0)
            mg.check(L.mgQueryReadsDeviceWait(t, C.byref(n), None))
            assert n.value == counts[i], ("seed count", i, n.value, counts[i])
        for i, b in enumerate(batches):                             # pipelined: results in slot 1
            t = C.c_void_p()
            mg.check(L.mgQueryReadsDeviceAsync(ms, b["p"].ptr, b["total"], b["o"].ptr, b["n"], b["ix"][1].ptr, b["pos"][1].ptr, b["rid"][1].ptr, b["cap"], C.byref(t), None))
            tickets.append((i, t))
            if len(tickets) == 2:
                wait_one()
        while tickets:
            wait_one()
        for i, b in enumerate(batches):
            for key in ("ix", "pos", "rid"):
                assert np.array_equal(b[key][0].to_numpy(np.uint32, counts[i]), b[key][1].to_numpy(np.uint32, counts[i])), (key, i)
        L.modsetDestroy(ms)


def main():
    t_end = time.time() + budget
    trials = 0
    have_min = hasattr(mg, "minimizer_batch")
    with tempfile.TemporaryDirectory() as td:
        import pathlib
        tmp = pathlib.Path(td)
        while time.time() < t_end:
            s = seed0 + trials
            rng = np.random.default_rng(s)
            k, w, sd = draw_params(rng)
            which = ["scan", "build", "query", "minimizer", "modmap", "readset", "pipelined"][trials % 7]
            if LARGE:
                which = "build"
                if k < 12:
                    k = int(rng.choice([15, 17, 19, 21, 25, 31]))      # (short k-mers: a handful of distinct values, nothing for the table to do)
            if which == "minimizer" and not have_min:
                which = "scan"
            t0 = time.time()
            try:
                if which == "scan":
                    trial_scan(rng, k, w, sd)
                elif which == "build":
                    trial_build(rng, k, w, sd)
                elif which == "query":
                    trial_query(rng, k, w, sd)
                elif which == "minimizer":
                    trial_minimizer(rng, k, w, sd)
                elif which == "pipelined":
                    trial_pipelined(rng, k, w, sd)
                else:
                    sub = tmp / ("t%d" % trials); sub.mkdir()
                    (trial_modmap if which == "modmap" else trial_readset)(rng, k, w, sd, sub)
                    for fn in os.listdir(sub):
                        os.remove(sub / fn)
                    sub.rmdir()
            except Exception:
                traceback.print_exc()
                print("SOAK_FAILED trial %d seed %d: %s k=%d w=%d hasher seed %d" % (trials, s, which, k, w, sd), flush=True)
                return 1
            print("trial %4d seed %8d %-9s k=%2d w=%5d seed %3d  %.1f s" % (trials, s, which, k, w, sd, time.time() - t0), flush=True)
            trials += 1
    print("SOAK_OK %d trials, first seed %d" % (trials, seed0), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
