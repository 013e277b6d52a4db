"""The oracle (oracle/*.c, this repo's CPU restatement) pinned against the reference:
 - the known answers captured from the compiled reference in SURVEY §8(c),
 - the committed golden vectors (tests/golden/, produced by running the reference here),
 - and, where oracle/_ref/libmodref.so is present, the reference library itself on fresh inputs.
"""
import os

import numpy as np
import pytest

from modimizer_amd import fasta, synth
from oracle import pyoracle as po
import util


def test_factor_constants():
    # glibc random() after srandom(17)
    h = po.Hasher(21, 64, 17)
    assert h.c.factor1 == 0x49308bb9003cb3ad and h.c.factor2 == 0x0fb4e87f75655103
    assert h.c.mask == 0x3ffffffffff and h.c.shift1 == 22
    assert po.Hasher(31, 4, 17).c.shift1 == 2


def test_known_answers_survey():
    r0, _ = synth.xorshift_read(10000)
    assert "".join("ACGT"[b] for b in r0[:31]) == "TCCAAGGTTGAACAGTAGTGCGAGAATCACC"
    k, p, f = po.Hasher(21, 64, 17).scan(r0)
    assert len(k) == 150
    assert [(int(p[i]), int(k[i]), int(f[i])) for i in range(6)] == [
        (1, 0x142be04b2e6, 1), (102, 0x3f9f59c97da, 0), (125, 0x3483585ba75, 0),
        (228, 0x1713d59d9fc, 0), (367, 0x6bad9e3a07, 1), (440, 0x382031b8643, 0)]
    h = po.Hasher(21, 64, 17)
    assert h.hash(0x142be04b2e6) == 0x130bce11e80
    k, p, f = po.Hasher(31, 4, 17).scan(r0)
    assert len(k) == 2409
    assert [(int(p[i]), int(k[i]), int(f[i])) for i in range(4)] == [
        (1, 0x0ae3dd91c7bd05fa, 0), (3, 0x02be04b2e620d17a, 1), (21, 0x220d17ac66102775, 1), (29, 0x06672889fed9b14a, 0)]
    k, p, f = po.Hasher(19, 31, 17).scan(r0)
    assert len(k) == 316
    assert [(int(p[i]), int(k[i]), int(f[i])) for i in range(3)] == [
        (6, 0x2f812cb988, 1), (86, 0x2e6b397896, 1), (115, 0x1ba759fe7d, 0)]


def test_known_answers_whole_file():
    # SURVEY §8(c), measured on the compiled reference: 1000 x 10 kb -> 155 837 hashes, max 155 837;
    # 10 000 x 10 kb -> 1 560 089 hashes, max 1 560 057, XOR over (kmer+pos) = 0x14abcb02167
    h = po.Hasher(21, 64, 17)
    ms = po.Modset(h, 24)
    bases, _ = po.xorshift_bases(10000 * 10000)
    assert np.array_equal(bases[:3000], synth.xorshift_read(3000)[0])
    total, x = 0, 0
    for r in range(10000):
        k, p, _ = h.scan(bases[r * 10000:(r + 1) * 10000])
        x ^= int(np.bitwise_xor.reduce(k + p.astype(np.uint64))) if len(k) else 0
        total += ms.add_sequence(bases[r * 10000:(r + 1) * 10000])
        if r == 999:
            assert total == 155837 and ms.max == 155837
    assert total == 1560089 and ms.max == 1560057 and x == 0x14abcb02167
    # k=31, d=4 on the same file: 24 925 287 hashes, XOR 0x193b2add2328e151
    h2 = po.Hasher(31, 4, 17)
    total, x = 0, 0
    for r in range(10000):
        k, p, _ = h2.scan(bases[r * 10000:(r + 1) * 10000])
        x ^= int(np.bitwise_xor.reduce(k + p.astype(np.uint64)))
        total += len(k)
    assert total == 24925287 and x == 0x193b2add2328e151


@pytest.mark.parametrize("ci", range(9))
def test_golden_scan_vectors(ci):
    k, w, seed = util.scan_configs()[ci]
    h = po.Hasher(k, w, seed)
    v = util.scan_vectors()
    assert h.c.factor1 == int(v["c%d_factor1" % ci][0]) and h.c.factor2 == int(v["c%d_factor1" % ci][1])
    for name, bases, kmer, pos, isf in util.scan_cases(ci):
        a, b, c = h.scan(bases)
        assert np.array_equal(a, kmer) and np.array_equal(b, pos) and np.array_equal(c, isf), (ci, name)
        m = util.minimizer_case(ci, name)
        if m is not None:
            x, y, z = h.minimizers(bases)
            assert np.array_equal(x, m[0]) and np.array_equal(y, m[1]) and np.array_equal(z, m[2]), (ci, name)


@pytest.mark.parametrize("tag", list(util.MODUTILS_TAGS))
def test_golden_modutils(tag, golden_dir, tmp_path):
    """modutils -c B k w s -a reads.fa -a reads2.fa -wt .. -H .. -p 2 40 -H .. -wt .. restated"""
    B, k, w, s = util.MODUTILS_TAGS[tag]
    h = po.Hasher(k, w, s)
    ms = po.Modset(h, B)
    tmp = str(tmp_path / "t.txt")
    out = "SH k %d  w/m %d  s %d\n" % (k, w, s)
    for fn in ("reads.fa", "reads2.fa"):
        names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, fn))
        tot = sum(ms.add_sequence(bases[offs[r]:offs[r + 1]]) for r in range(len(names)))
        out += "added %d sequences total length %d total hashes %d, new max %d\n" % (len(names), offs[-1], tot, ms.max)
        out += ms.summary_text(tmp)
    util.check_dump(ms.text_dump(tmp), "modutils_%s.dump.txt" % tag)
    assert ms.hist_text(tmp) == util.golden_text("modutils_%s.hist.txt" % tag)
    ms.prune(2, 40)
    out += ms.summary_text(tmp)
    assert ms.hist_text(tmp) == util.golden_text("modutils_%s.pruned_hist.txt" % tag)
    util.check_dump(ms.text_dump(tmp), "modutils_%s.pruned_dump.txt" % tag)
    assert out == util.golden_text("modutils_%s.stdout.txt" % tag)


@pytest.mark.parametrize("tag", list(util.MODMAP_TAGS))
def test_golden_modmap(tag, golden_dir, tmp_path):
    """modmap -K k -W w -S 17 -B 20 -f ref.fa -q queries.fa restated"""
    k, w = util.MODMAP_TAGS[tag]
    h = po.Hasher(k, w, 17)
    ms = po.Modset(h, 20)
    ref = po.Reference(ms, 1 << 26)
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "ref.fa"))
    for r, n in enumerate(names):
        ref.add_sequence(n, bases[offs[r]:offs[r + 1]])
    ref.finish()
    a = ref.arrays()
    out = "  modmap initialised with k = %d, w = %d, random seed = 17\n" % (k, w)
    out += "  %d hashes from %d reference sequences, total length %d\n" % (len(a["index"]), len(names), offs[-1])
    out += "  %d copy 1, %d copy 2, %d multiple\n" % (a["n1"], a["n2"], a["nM"])
    qn, qb, qo = fasta.read_fasta(os.path.join(golden_dir, "queries.fa"))
    tmp = str(tmp_path / "q.txt")
    for r, n in enumerate(qn):
        txt, _, _ = ref.query(n, qb[qo[r]:qo[r + 1]], tmp)
        out += txt
    assert out == util.golden_text("modmap_%s.stdout.txt" % tag)
    ref.close()


def _load_ops():
    return np.load(os.path.join(util.GOLDEN, "modset_ops.npz"))


def _build_ab(golden_dir):
    g = _load_ops()
    k, w, seed, B = (int(x) for x in g["params"])
    h = po.Hasher(k, w, seed)
    a, b = po.Modset(h, B), po.Modset(h, B)
    n1, b1, o1 = fasta.read_fasta(os.path.join(golden_dir, "reads.fa"))
    n2, b2, o2 = fasta.read_fasta(os.path.join(golden_dir, "reads2.fa"))
    for r in range(len(n1)):
        a.add_sequence(b1[o1[r]:o1[r + 1]])
    for r in range(len(n2)):
        b.add_sequence(b2[o2[r]:o2[r + 1]])
    for i in range(1, b.max + 1):
        b.p.contents.info[i] = (i % 4) | ((i % 3 == 0) * 8)
    for i in range(1, a.max + 1):
        a.p.contents.info[i] = ((i // 2) % 4) | ((i % 5 == 0) * 16)
    return g, a, b, B


def _same(ms, g, tag, B, tmp):
    assert np.array_equal(ms.values()[1:], g[tag + "_value"][1:])      # value[0] is uninitialised in the reference
    assert np.array_equal(ms.depths(), g[tag + "_depth"])
    assert np.array_equal(ms.infos(), g[tag + "_info"])
    idx = ms.index_table()
    nz = np.nonzero(idx)[0]
    assert np.array_equal(nz.astype(np.uint32), g[tag + "_index_pos"]) and np.array_equal(idx[nz], g[tag + "_index_val"])
    assert ms.summary_text(tmp).encode() == g[tag + "_summary"].tobytes()


def test_golden_modset_ops(golden_dir, tmp_path):
    import hashlib
    g, a, b, B = _build_ab(golden_dir)
    tmp = str(tmp_path / "s.txt")
    _same(a, g, "a", B, tmp)
    _same(b, g, "b", B, tmp)
    mod = str(tmp_path / "a.mod")
    a.p.contents.value[0] = 0
    a.write_mod(mod)
    data = open(mod, "rb").read()
    assert len(data) == int(g["a_mod_len"][0]) == 104 + 4 * (1 << B) + 11 * (a.max + 1)
    assert data[:104] == g["a_mod_header"].tobytes()
    assert hashlib.sha256(data).digest() == g["a_mod_sha256"].tobytes()
    assert a.merge(b)
    _same(a, g, "merged", B, tmp)
    a.prune(2, 30)
    _same(a, g, "pruned", B, tmp)
    a.pack()
    assert a.p.contents.size == int(g["packed_size"][0])


needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("k,w,seed", [(21, 64, 17), (31, 4, 17), (19, 31, 17), (16, 32, 0), (12, 7, 99), (1, 2, 3), (31, 1, 5)])
def test_vs_reference_random(k, w, seed):
    R = po.ref()
    sh = R.seqhashCreate(k, w, seed)
    h = po.Hasher(k, w, seed)
    assert sh.contents.factor1 == h.c.factor1 and sh.contents.factor2 == h.c.factor2
    rng = np.random.default_rng(k * 131 + w)
    rms = R.modsetCreate(sh, 20, 0)
    oms = po.Modset(h, 20)
    for L in [0, 1, k - 1, k, k + 1, 2 * k, 257, 1000, 4000]:
        b = rng.integers(0, 4, L).astype(np.uint8)
        if L > 100:
            b[L // 3:L // 3 + 50] = b[10:60]        # a repeat, so some k-mers recur
        a, c = h.scan(b), po.ref_scan(sh, b)
        assert all(np.array_equal(x, y) for x, y in zip(a, c)), (k, w, L)
        if L >= k and w <= 64:
            a, c = h.minimizers(b), po.ref_scan(sh, b, minimizer=True)
            assert all(np.array_equal(x, y) for x, y in zip(a, c)), ("minimizer", k, w, L)
        if w > 1 or L < 2000:
            assert po.ref_add_sequence(rms, b) == oms.add_sequence(b)
    n = rms.contents.max
    assert n == oms.max
    assert np.array_equal(np.ctypeslib.as_array(rms.contents.value, (n + 1,))[1:], oms.values()[1:])
    assert np.array_equal(np.ctypeslib.as_array(rms.contents.depth, (n + 1,)), oms.depths())
    assert np.array_equal(np.ctypeslib.as_array(rms.contents.index, (1 << 20,)), oms.index_table())
    for km in list(oms.values()[1:20]) + [12345, 0]:
        assert R.modsetIndexFind(rms, int(km), 0) == oms.find(km)


@needs_ref
def test_vs_reference_depth_saturation():
    """modutils.c:26: the U16 depth wraps to 0 and is pinned to 65535"""
    R = po.ref()
    k, w = 3, 1
    sh = R.seqhashCreate(k, w, 17); h = po.Hasher(k, w, 17)
    b = np.zeros(70000, np.uint8)
    rms = R.modsetCreate(sh, 20, 0); oms = po.Modset(h, 20)
    po.ref_add_sequence(rms, b); oms.add_sequence(b)
    assert oms.max == 1 and rms.contents.max == 1
    assert oms.depths()[1] == 65535 == rms.contents.depth[1]


def test_first_occurrences_helper_matches_numpy():
    """oracle/orc_modset.c orcFirstOccurrences (the full-size parity tests' host-side reconstruction of value[] / depth[])
    against the sort-based definition and against sequential orcModsetFind inserts"""
    import ctypes as C
    sys_path_fullsize = os.path.join(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location("fullsize_whole", os.path.join(sys_path_fullsize, "fullsize_whole.py"))
    fw = importlib.util.module_from_spec(spec); spec.loader.exec_module(fw)
    rng = np.random.default_rng(5)
    for n, distinct in ((0, 1), (1, 1), (1000, 7), (300_000, 40_000), (300_000, 10**12)):
        km = (rng.integers(0, distinct, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64((1 << 42) - 1)
        v, d = fw.first_occurrence_arrays(km)
        uq, fi, ct = np.unique(km, return_index=True, return_counts=True)
        order = np.argsort(fi, kind="stable")
        assert np.array_equal(v, uq[order]) and np.array_equal(d, np.minimum(ct[order], 65535).astype(np.uint16))
    # saturation, and the oracle's own modset on the same stream
    km = np.concatenate([np.full(70_000, 5, np.uint64), rng.integers(0, 50, 5000).astype(np.uint64)])
    v, d = fw.first_occurrence_arrays(km)
    assert d[0] == 65535
    oh = po.Hasher(21, 64, 17); oms = po.Modset(oh, 20)
    for x in km:
        ix = oms.find(int(x), True)
    assert oms.max == len(v) and np.array_equal(oms.values()[1:], v)


def test_first_occurrence_positions_and_reference_pack_vs_numpy():
    """the two helpers the full-size reference test rebuilds modmap's arrays with (tests/fullsize_whole.py c3ref; round 6: 66 s of numpy sorts
    -> a few seconds): orcFirstOccurrencesAt's first-occurrence position of every k-mer of a stream, and orcReferencePack = referencePack's own
    loops (modmap.c:74-91), against numpy's sort-based statement of the same (np.unique, stable argsort, bincount, cumsum)"""
    import ctypes as C
    OL = po.lib()
    rng = np.random.default_rng(3)
    for n, universe in ((200_000, 5000), (50_000, 1 << 40), (1, 1)):
        km = rng.integers(0, universe, n).astype(np.uint64)
        OL.orcFirstOccurrencesAt.restype = C.c_int64
        OL.orcFirstOccurrencesAt.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        flag = np.zeros(n, np.uint8); cnt = np.zeros(n, np.uint32); fa = np.zeros(n, np.uint32)
        u = OL.orcFirstOccurrencesAt(km.ctypes.data, n, 4, flag.ctypes.data, cnt.ctypes.data, fa.ctypes.data)
        vals, first_idx, counts = np.unique(km, return_index=True, return_counts=True)
        assert u == len(vals) and np.array_equal(np.flatnonzero(flag), np.sort(first_idx))
        assert np.array_equal(fa, first_idx[np.searchsorted(vals, km)].astype(np.uint32))
        assert np.array_equal(cnt[first_idx], counts.astype(np.uint32))
        occ_index = np.cumsum(flag, dtype=np.uint32)[fa]
        want_value = km[flag.view(bool)]
        order = np.argsort(want_value, kind="stable")
        assert np.array_equal((order[np.searchsorted(want_value[order], km)] + 1).astype(np.uint32), occ_index)
        U = int(u)
        depth = np.zeros(U + 1, np.uint32); loc = np.zeros(U + 1, np.uint32); rev = np.zeros(n, np.uint32)
        OL.orcReferencePack.restype = None
        OL.orcReferencePack.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        OL.orcReferencePack(occ_index.ctypes.data, n, U, depth.ctypes.data, loc.ctypes.data, rev.ctypes.data)
        assert np.array_equal(rev, np.argsort(occ_index, kind="stable").astype(np.uint32))
        assert np.array_equal(depth, np.bincount(occ_index, minlength=U + 1).astype(np.uint32))
        l2 = np.zeros(U + 1, np.uint32); l2[1:] = np.cumsum(depth[:-1], dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(loc, l2)
