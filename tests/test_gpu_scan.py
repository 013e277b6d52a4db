"""GPU parity: the HIP scan (K1/K2) through the C ABI vs the oracle and the golden vectors.
Bit-exact: these are integer k-mers, positions and strand flags."""
import ctypes as C

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import synth
from oracle import pyoracle as po
import util

pytestmark = pytest.mark.gpu
TILE = 4096            # MG_TILE_BASES (mg_common.h): k-mer starts per tile = one wavefront-worker's unit
DEFAULT_BELOW = mg.lib().mgIterHostBelow(-1)     # modRCiterator's crossover in force at start-up (mg_host.c)


def assert_batch_equal(sh, oh, reads):
    bases, offs = util.concat_reads(reads)
    km, pos, isf, st = mg.scan_batch(sh, bases, offs)
    ek, ep, ef, est = util.oracle_scan_batch(oh, bases, offs)
    assert np.array_equal(st, est)
    assert np.array_equal(km, ek) and np.array_equal(pos, ep) and np.array_equal(isf, ef)
    return len(km)


@pytest.mark.parametrize("ci", range(9))
def test_golden_vectors_batch_and_iterator(ci):
    """every golden read of a config in ONE batch (ragged, incl. empty / len<k / len==k reads)"""
    k, w, seed = util.scan_configs()[ci]
    sh = mg.seqhashCreate(k, w, seed)
    cases = list(util.scan_cases(ci))
    bases, offs = util.concat_reads([c[1] for c in cases])
    km, pos, isf, st = mg.scan_batch(sh, bases, offs)
    for r, (name, b, gk, gp, gf) in enumerate(cases):
        s, e = st[r], st[r + 1]
        assert np.array_equal(km[s:e], gk) and np.array_equal(pos[s:e], gp) and np.array_equal(isf[s:e], gf), (ci, name)
        # the per-read facade the reference callers use (modRCiterator/modRCnext), both legs: every read through the
        # kernel (crossover 0) and short reads through the scalar loop (the default crossover)
        for below in (0, 1 << 30):
            mg.lib().mgIterHostBelow(below)                                 # 1 << 30: the defaults
            a, p, f = mg.iterate(sh, b)
            assert np.array_equal(a, gk) and np.array_equal(p, gp) and np.array_equal(f, gf), (ci, name, "iterator", below)


@pytest.mark.parametrize("k,w,seed", [(21, 64, 17), (31, 4, 17), (19, 31, 17), (16, 32, 0), (11, 1, 3),
                                      (1, 1, 17), (2, 3, 5), (31, 97, 9), (27, 1024, 17), (5, 2, 17),
                                      (21, 96, 17), (13, 6, 5), (31, 12, 2), (17, 1000, 1), (16, 31, 4), (15, 7, 4)])   # even d that is no power of two (MG_MODE_ANY), odd d around k = 16
def test_random_ragged_batches(k, w, seed):
    sh = mg.seqhashCreate(k, w, seed); oh = po.Hasher(k, w, seed)
    rng = np.random.default_rng(k * 7 + w)
    lens = [0, 1, k - 1, k, k + 1, 63, 64, 65, 127, 128, 1000, 0, 0, 5, 4097, 10000, 2, 31, 32, 33, 64 + k - 1, 64 + k]
    reads = [rng.integers(0, 4, L).astype(np.uint8) for L in lens]
    n = assert_batch_equal(sh, oh, reads)
    assert n > 0


def test_tile_boundaries():
    """reads that start / end exactly at, just before and just after the tile edges (TILE bases), and whose last
    k-mer (start len-k) falls on either side of an edge: TILE-1, TILE, TILE+1, TILE+k-1 and neighbours"""
    k, w = 21, 8
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(1)
    lens = [TILE - k + 1, k - 1, TILE, 1, TILE - 1, 2, TILE + k - 1, TILE - 20, 20, 21, 22, 3 * TILE + 5, 19, TILE // 2, TILE // 2,
            TILE + 1, TILE - 1, TILE + k, TILE + k - 2, 2 * TILE - k, 2 * TILE + k - 1, k, TILE - k, TILE - k - 1, TILE - k + 2]
    reads = [rng.integers(0, 4, L).astype(np.uint8) for L in lens]
    assert_batch_equal(sh, oh, reads)
    # a batch that ends exactly on a tile edge, and one base short / over; one read per case
    for total in (TILE, TILE - 1, TILE + 1, TILE + k - 1, TILE + k - 2, TILE + k, 2 * TILE, 64, 63, 65, 4 * TILE + k - 1):
        assert_batch_equal(sh, oh, [rng.integers(0, 4, total).astype(np.uint8)])
    # every read boundary position around one tile edge, with the d = 1 hasher (every start is a modimizer) and with
    # the filtering kernel (k=21 d=64) on a read pair swept across the edge
    for kk, ww in ((5, 1), (21, 64), (31, 4)):
        sh2 = mg.seqhashCreate(kk, ww, 17); oh2 = po.Hasher(kk, ww, 17)
        body = rng.integers(0, 4, 3 * TILE).astype(np.uint8)
        for cut in list(range(TILE - kk - 1, TILE + kk + 2)) + [2 * TILE - 1, 2 * TILE, 2 * TILE + 1]:
            assert_batch_equal(sh2, oh2, [body[:cut], body[cut:]])


WORKER_RANGE_CODE = r"""
import numpy as np, sys
sys.path.insert(0, "tests")
import modimizer_amd as mg
from oracle import pyoracle as po
import util
TILE = 4096
rng = np.random.default_rng(7)
for k, w in ((21, 64), (21, 8), (31, 4), (19, 31)):
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    # 3 workers over 11+ tiles: ranges of 4 tiles; reads ending exactly at, one before and one after the range edges
    # (4*TILE, 8*TILE), reads spanning a range edge, and a run of short reads across one
    lens = [4 * TILE, 4 * TILE - 1, 1, 4 * TILE + k - 1, TILE - k + 1, 3 * TILE]
    for reads in ([rng.integers(0, 4, L).astype(np.uint8) for L in lens],
                  [rng.integers(0, 4, 4 * TILE - 100).astype(np.uint8)] + [rng.integers(0, 4, int(L)).astype(np.uint8) for L in rng.integers(0, 60, 40)]
                  + [rng.integers(0, 4, 7 * TILE).astype(np.uint8)],
                  [rng.integers(0, 4, 12 * TILE + 17).astype(np.uint8)]):
        bases, offs = util.concat_reads(reads)
        km, pos, isf, st = mg.scan_batch(sh, bases, offs)
        ek, ep, ef, est = util.oracle_scan_batch(oh, bases, offs)
        assert np.array_equal(st, est) and np.array_equal(km, ek) and np.array_equal(pos, ep) and np.array_equal(isf, ef), (k, w)
print("ok")
"""


def test_worker_range_boundaries():
    """MODGPU_SCAN_GRID=3: three wavefront-workers, each owning a RANGE of consecutive tiles and its own output segment;
    reads that end on, straddle and crowd the range edges (a small batch otherwise gets one tile per worker)"""
    import os, subprocess, sys
    r = subprocess.run([sys.executable, "-c", WORKER_RANGE_CODE], capture_output=True, text=True, cwd=util.ROOT,
                       env=dict(os.environ, MODGPU_SCAN_GRID="3"), timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_many_short_reads():
    """Illumina-like: 30 000 x 150 b, k=31 d=4 (config 5's shape): ~110 read boundaries per tile"""
    k, w = 31, 4
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    genome = synth.iid_bases(200000, 3)
    starts, offs, strands = synth.fixed_read_plan(30000, 150, len(genome), 4)
    bases = synth.reads_from_genome(genome, starts, offs, strands, 0.005, 5)
    km, pos, isf, st = mg.scan_batch(sh, bases, offs.astype(np.int64))
    ek, ep, ef, est = util.oracle_scan_batch(oh, bases, offs.astype(np.int64))
    assert np.array_equal(st, est) and np.array_equal(km, ek) and np.array_equal(pos, ep) and np.array_equal(isf, ef)


def test_tiny_reads_stress():
    """thousands of reads shorter than, equal to and slightly longer than k inside single tiles"""
    k, w = 7, 3
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(9)
    reads = [rng.integers(0, 4, int(L)).astype(np.uint8) for L in rng.integers(0, 20, 5000)]
    assert_batch_equal(sh, oh, reads)


def test_one_long_sequence():
    """a 3 Mbp 'chromosome': the scan must not depend on one-wave-per-read"""
    k, w = 21, 64
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b = synth.iid_bases(3_000_000, 77)
    n = assert_batch_equal(sh, oh, [b])
    assert abs(n - 3_000_000 / 64) < 2000


def test_palindromes_and_homopolymers():
    """hashF == hashR ties resolve to the reverse strand (seqhash.c:66-67)"""
    for k, w in [(16, 1), (4, 1), (2, 1), (16, 2)]:
        sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
        reads = [np.array([0, 1, 2, 3] * 50, np.uint8), np.zeros(200, np.uint8), np.full(200, 3, np.uint8),
                 np.array([0, 3] * 100, np.uint8), np.array([1, 2] * 100, np.uint8)]
        assert_batch_equal(sh, oh, reads)
        km, pos, isf, st = mg.scan_batch(sh, *util.concat_reads(reads))
        if w == 1:
            assert (isf[st[0]:st[1]] == 0).any()      # palindromic k-mers are reported as reverse


def test_device_api_capacity_and_count():
    """seqhashScanBatchDevice: dCount = {true total, overflow flag, fullest segment, retry capacity};
    a retry with dCount[3] always succeeds"""
    L = mg.lib()
    k, w = 21, 16
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(3)
    # deliberately skewed: a poly-A stretch whose every k-mer is the same modimizer-or-not, then random
    reads = [rng.integers(0, 4, L_).astype(np.uint8) for L_ in (40000, 123, 70000, 9)]
    bases, offs = util.concat_reads(reads)
    ek, ep, ef, _ = util.oracle_scan_batch(oh, bases, offs)
    total = len(bases)
    d_packed = mg.DeviceBuffer.from_numpy(mg.pack_host(bases))
    d_off = mg.DeviceBuffer.from_numpy(offs.astype(np.uint64))
    d_cnt = mg.DeviceBuffer(32)

    def run(cap, with_rid=True):
        d_work = mg.DeviceBuffer(L.mgScanWorkBytes(total, len(reads), cap))
        d_k = mg.DeviceBuffer(max(cap, 1) * 8); d_p = mg.DeviceBuffer(max(cap, 1) * 4); d_r = mg.DeviceBuffer(max(cap, 1) * 4)
        mg.check(L.seqhashScanBatchDevice(sh, d_packed.ptr, total, d_off.ptr, len(reads), d_k.ptr, d_p.ptr,
                                          d_r.ptr if with_rid else None, cap, d_cnt.ptr, d_work.ptr, None))
        return d_cnt.to_numpy(np.uint64, 4), d_k, d_p, d_r
    for cap in (2 * len(ek), len(ek) + len(ek) // 4, len(ek), 100, 0):
        cnt, d_k, d_p, d_r = run(cap)
        assert cnt[0] == len(ek)
        if cap < len(ek):
            assert cnt[1] == 1
        if cnt[1]:
            assert cnt[3] >= len(ek)
            cnt, d_k, d_p, d_r = run(int(cnt[3]))
            assert cnt[0] == len(ek) and cnt[1] == 0
        m = len(ek)
        assert np.array_equal(d_k.to_numpy(np.uint64, m), ek)
        pf = d_p.to_numpy(np.uint32, m)
        assert np.array_equal(pf & mg.MG_POS_MASK, ep.astype(np.uint32)) and np.array_equal((pf >> 31).astype(np.uint8), ef)
        rid = d_r.to_numpy(np.uint32, m)
        assert (np.diff(rid.astype(np.int64)) >= 0).all()
    # dReadId may be NULL
    cnt, d_k, d_p, _ = run(2 * len(ek), with_rid=False)
    assert cnt[1] == 0 and np.array_equal(d_k.to_numpy(np.uint64, len(ek)), ek)


def test_skewed_density_retries():
    """all modimizers concentrated in one region (one workgroup's segment): the host entry points retry"""
    k, w = 21, 64
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(11)
    # find a 21-mer that IS a modimizer and tile it: every k-mer start hits
    b = rng.integers(0, 4, 200000).astype(np.uint8)
    km, pos, _ = oh.scan(b)
    unit = b[pos[0]:pos[0] + k]
    rep = np.tile(unit, 3000)                      # 63 kb where a hit recurs every k bases (period k)
    reads = [rng.integers(0, 4, 3_000_000).astype(np.uint8), rep, rng.integers(0, 4, 1_000_000).astype(np.uint8)]
    assert_batch_equal(sh, oh, reads)


def test_pack_unpack_and_synth_on_device():
    L = mg.lib()
    rng = np.random.default_rng(5)
    for n in (0, 1, 15, 16, 17, 1000, 100003):
        b = rng.integers(0, 4, n).astype(np.uint8)
        d_b = mg.DeviceBuffer.from_numpy(b)
        d_w = mg.DeviceBuffer(L.mgPackedWords(n) * 4)
        mg.check(L.mgPackDevice(d_b.ptr, n, d_w.ptr, None))
        assert np.array_equal(d_w.to_numpy(np.uint32, L.mgPackedWords(n)), mg.pack_host(b))
        d_o = mg.DeviceBuffer(max(n, 1))
        mg.check(L.mgUnpackDevice(d_w.ptr, n, d_o.ptr, None))
        assert np.array_equal(d_o.to_numpy(np.uint8, n), b)
    # device generators == their numpy mirrors (the bench's inputs are reproducible on the host)
    G = 50001
    d_g = mg.DeviceBuffer(L.mgPackedWords(G) * 4)
    mg.check(L.mgSynthGenome(d_g.ptr, G, 12345, None))
    genome = synth.iid_bases(G, 12345)
    assert np.array_equal(d_g.to_numpy(np.uint32, L.mgPackedWords(G)), mg.pack_host(genome))
    starts, offs, strands = synth.ont_read_plan(200000, G, 7, n50=3000, lo=100, hi=20000)
    total = int(offs[-1])
    d_s = mg.DeviceBuffer.from_numpy(starts); d_of = mg.DeviceBuffer.from_numpy(offs); d_st = mg.DeviceBuffer.from_numpy(strands)
    d_r = mg.DeviceBuffer(L.mgPackedWords(total) * 4)
    mg.check(L.mgSynthReads(d_g.ptr, G, d_s.ptr, d_of.ptr, d_st.ptr, len(starts), total, 0.05, 99, d_r.ptr, None))
    host = synth.reads_from_genome(genome, starts, offs, strands, 0.05, 99)
    assert np.array_equal(d_r.to_numpy(np.uint32, L.mgPackedWords(total)), mg.pack_host(host))


def test_full_size_properties():
    """1 Gbp of device-generated ONT-like reads (BASELINE config 2's shape at 1/10 scale, then the
    full 10 Gbp): properties that need no oracle — splitting the batch at a read boundary and
    scanning the halves gives the same multiset checksum and count (the scan is per-read);
    positions are increasing within a read; read ids are non-decreasing."""
    L = mg.lib()
    sh = mg.seqhashCreate(21, 64, 17)
    for gbp in (1.0, 10.0):
        total = int(gbp * 1e9)
        G = total // 30
        starts, offs, strands = synth.ont_read_plan(total, G, 1000)
        n_reads = len(starts)
        d_g = mg.DeviceBuffer(L.mgPackedWords(G) * 4)
        mg.check(L.mgSynthGenome(d_g.ptr, G, 12345, None))
        d_s = mg.DeviceBuffer.from_numpy(starts); d_of = mg.DeviceBuffer.from_numpy(offs); d_st = mg.DeviceBuffer.from_numpy(strands)
        d_r = mg.DeviceBuffer(L.mgPackedWords(total) * 4)
        mg.check(L.mgSynthReads(d_g.ptr, G, d_s.ptr, d_of.ptr, d_st.ptr, n_reads, total, 0.05, 777, d_r.ptr, None))
        d_g.free()
        cap = int(total / 64 * 1.1)
        d_k = mg.DeviceBuffer(cap * 8); d_p = mg.DeviceBuffer(cap * 4); d_id = mg.DeviceBuffer(cap * 4)
        d_cnt = mg.DeviceBuffer(32); d_work = mg.DeviceBuffer(L.mgScanWorkBytes(total, n_reads, cap))

        def run(off_arr, nr, tot, packed_ptr):
            d_o = mg.DeviceBuffer.from_numpy(off_arr)
            mg.check(L.seqhashScanBatchDevice(sh, packed_ptr, tot, d_o.ptr, nr, d_k.ptr, d_p.ptr, d_id.ptr, cap, d_cnt.ptr, d_work.ptr, None))
            c = d_cnt.to_numpy(np.uint64, 4)
            assert c[1] == 0
            n = int(c[0])
            km = d_k.to_numpy(np.uint64, n); pf = d_p.to_numpy(np.uint32, n); rid = d_id.to_numpy(np.uint32, n)
            return km, pf, rid
        km, pf, rid = run(offs, n_reads, total, d_r.ptr)
        assert abs(len(km) / (total / 64.0) - 1) < 0.01              # 1-in-d density
        assert (np.diff(rid.astype(np.int64)) >= 0).all()
        same = rid[1:] == rid[:-1]
        assert ((pf[1:] & mg.MG_POS_MASK)[same] > (pf[:-1] & mg.MG_POS_MASK)[same]).all()
        lens = (offs[1:] - offs[:-1]).astype(np.int64)
        assert ((pf & mg.MG_POS_MASK).astype(np.int64) <= lens[rid] - 21).all()
        full = (int(np.bitwise_xor.reduce(km + pf.astype(np.uint64))), len(km), int(km.sum(dtype=np.uint64)))
        # first half only: the reads before a 16-base-aligned boundary -> prefix of the full result
        cut = int(np.searchsorted(offs, total // 2))
        while cut < n_reads and int(offs[cut]) % 16:
            cut += 1
        if cut < n_reads:
            km1, pf1, rid1 = run(offs[:cut + 1].copy(), cut, int(offs[cut]), d_r.ptr)
            n1 = len(km1)
            assert np.array_equal(km1, km[:n1]) and np.array_equal(pf1, pf[:n1]) and (rid[:n1] < cut).all() and rid[n1] >= cut
            off2 = (offs[cut:] - offs[cut]).copy()
            km2, pf2, rid2 = run(off2, n_reads - cut, total - int(offs[cut]), C.c_void_p(d_r.ptr.value + int(offs[cut]) // 4))
            assert np.array_equal(km2, km[n1:]) and np.array_equal(pf2, pf[n1:]) and np.array_equal(rid2 + cut, rid[n1:])
        del km, pf, rid
        for b in (d_k, d_p, d_id, d_r):
            b.free()
        assert full[1] > 0


def test_upload_pack_multi_piece():
    """mgUploadPack: host bytes -> HBM in 64 Mbase pieces, packed on the device == mgPackHost"""
    L = mg.lib()
    rng = np.random.default_rng(17)
    for n in (0, 1, 17, (64 << 20) - 3, (64 << 20) + 5, 150_000_001):
        b = rng.integers(0, 4, n, dtype=np.uint8)
        d_w = mg.DeviceBuffer(L.mgPackedWords(n) * 4)
        mg.check(L.mgUploadPack(b.ctypes.data, n, d_w.ptr, None))
        got = d_w.to_numpy(np.uint32, L.mgPackedWords(n))
        assert np.array_equal(got, mg.pack_host(b)), n


# ---- minimizers (seqhash.c:83-152): one wavefront per read walks the reference's chain of windows ----

@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(9))
def test_minimizer_iterator_vs_golden(ci):
    """minimizerRCiterator / minimizerRCnext against the lists the reference produced (edge reads: empty,
    len<k, len==k, homopolymers with all hashes equal, the hashBuf[0] case)"""
    k, w, seed = util.scan_configs()[ci]
    sh = mg.seqhashCreate(k, w, seed)
    for name, bases, _, _, _ in util.scan_cases(ci):
        m = util.minimizer_case(ci, name)
        if m is None:
            continue
        x, y, z = mg.iterate(sh, bases, minimizer=True)
        assert np.array_equal(x, m[0]) and np.array_equal(y, m[1]) and np.array_equal(z, m[2]), (ci, name)


@pytest.mark.gpu
@pytest.mark.parametrize("k,w,seed", [(21, 31, 17), (15, 5, 3), (31, 64, 17), (19, 100, 9), (12, 1, 2), (21, 2, 17),
                                      (21, 256, 17), (21, 257, 17), (15, 7, 3), (9, 3, 5), (19, 1023, 9)])
@pytest.mark.parametrize("tiled", [None, "0"])
def test_minimizer_batch_vs_oracle(k, w, seed, tiled):
    """a ragged batch (empty, shorter than k, shorter than k+w, monotone runs, homopolymers, long reads); through the tiles (windows
    up to 256: suffix / prefix arg-minima per block of w) and one link at a time (MODGPU_MIN_TILED=0, and every w > 256); k = 9 with
    w = 3 on 200 kb makes thousands of equal hashes per window: the tie rule (smallest position mod w) on every link"""
    with mg.knobs(MIN_TILED=tiled):
        _minimizer_batch_vs_oracle(k, w, seed)


def _minimizer_batch_vs_oracle(k, w, seed):
    from oracle import pyoracle as orc
    rng = np.random.default_rng(k * 1000 + w)
    reads = [np.zeros(0, np.uint8), rng.integers(0, 4, k - 1).astype(np.uint8), rng.integers(0, 4, k).astype(np.uint8),
             rng.integers(0, 4, k + 1).astype(np.uint8), rng.integers(0, 4, k + w - 2).astype(np.uint8),
             rng.integers(0, 4, k + w - 1).astype(np.uint8), rng.integers(0, 4, k + w).astype(np.uint8),
             np.zeros(5 * w + k + 3, np.uint8), np.full(3 * w + k, 3, np.uint8), np.tile(np.array([0, 1], np.uint8), 4 * w + k),
             np.tile(np.array([0, 1, 2, 3, 3, 2], np.uint8), 300)]
    reads += [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(k, 6000, 60)]
    reads += [rng.integers(0, 4, 200_000).astype(np.uint8)]
    h = orc.Hasher(k, w, seed)
    want = [h.minimizers(r) for r in reads]
    sh = mg.seqhashCreate(k, w, seed)
    bases, offs = util.concat_reads(reads)
    hv, pos, isf, st = mg.minimizer_batch(sh, bases, offs)
    assert st[0] == 0 and st[-1] == len(hv) == sum(len(x[0]) for x in want)
    for r, (a, b, c) in enumerate(want):
        s, e = st[r], st[r + 1]
        assert np.array_equal(hv[s:e], a) and np.array_equal(pos[s:e], b) and np.array_equal(isf[s:e], c), r


@pytest.mark.gpu
def test_minimizer_batch_capacity_is_reported():
    L = mg.lib()
    sh = mg.seqhashCreate(21, 11, 17)
    rng = np.random.default_rng(5)
    bases = rng.integers(0, 4, 50_000).astype(np.uint8)
    offs = np.array([0, 20_000, 50_000], np.int64)
    hv, pos, isf, st = mg.minimizer_batch(sh, bases, offs)
    n_true = len(hv)
    packed = mg.pack_host(bases)
    d_p = mg.DeviceBuffer.from_numpy(packed)
    d_o = mg.DeviceBuffer.from_numpy(offs.astype(np.uint64))
    d_h = mg.DeviceBuffer(8 * 16); d_q = mg.DeviceBuffer(4 * 16); d_s = mg.DeviceBuffer(8 * 3)
    n = C.c_uint64()
    rc = L.seqhashMinimizerBatchDevice(sh, d_p.ptr, len(bases), d_o.ptr, 2, d_h.ptr, d_q.ptr, d_s.ptr, 16, C.byref(n), None)
    assert rc == 4 and n.value == n_true            # MG_ERR_CAPACITY
    assert list(d_s.to_numpy(np.uint64, 3)) == [0, int(st[1]), n_true]


@pytest.mark.gpu
def test_fuzz_small():
    """tests/fuzz_gpu.py: random (k, d, seed) and ragged batches (reads around tile edges, homopolymer stretches),
    scan / modset with its index[] layout / minimizers against the oracle"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_gpu.py"), "5", "120"], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr[-500:]
    assert "120 trials, 0 mismatches" in r.stdout, r.stdout[-500:]


def _iterate_arrays(sh, bases, below=0):
    """modRCiterator / modRCnext over one read without a Python loop per modimizer: the replay block behind hashBuf
    (mg_host.c) is {n, n k-mers, n words pos | isF << 31}"""
    L = mg.lib()
    L.mgIterHostBelow(below)                                                    # 0: every read through the kernel
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    it = L.modRCiterator(sh, bases.ctypes.data, len(bases))
    addr = it.contents.hashBuf                                                  # a void * in the binding
    n = int(C.cast(C.c_void_p(addr), C.POINTER(C.c_uint64))[0])
    km = np.ctypeslib.as_array(C.cast(C.c_void_p(addr + 8), C.POINTER(C.c_uint64)), (max(n, 1),))[:n].copy()
    pf = np.ctypeslib.as_array(C.cast(C.c_void_p(addr + 8 * (n + 1)), C.POINTER(C.c_uint32)), (max(n, 1),))[:n].copy()
    # and the public face of it, for the first few
    u = C.c_uint64(); p = C.c_int(); f = C.c_bool()
    for i in range(min(n, 5)):
        assert L.modRCnext(it, C.byref(u), C.byref(p), C.byref(f))
        assert u.value == int(km[i]) and p.value == int(pf[i] & mg.MG_POS_MASK) and bool(f.value) == bool(pf[i] >> 31)
    L.mgSeqhashRCiteratorDestroy(it)
    L.mgIterHostBelow(1 << 30)                                                  # back to the defaults by w
    return km, (pf & np.uint32(mg.MG_POS_MASK)).astype(np.int32), (pf >> 31).astype(np.uint8)


@pytest.mark.parametrize("k,w", [(21, 64), (31, 4), (19, 31), (5, 1), (17, 8)])
def test_iterator_one_launch_kernel_edges(k, w):
    """modRCiterator (seqhash.c:154-196) = ONE kernel launch per read up to 64 tiles (mgIterScanKernel: eight workers, the
    replay block written into pinned host memory), the batch scan beyond: lengths around k, around the tile edge, around the
    points where the number of tiles per worker changes (8 and 16 tiles), at the one-launch limit and past it; w = 1 makes
    every start a modimizer, so the first launch's output estimate is too small and the call retries with the size the
    kernel reported."""
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(k + w)
    lens = [0, 1, k - 1, k, k + 1, 100, TILE - 1, TILE, TILE + 1, TILE + k - 2, TILE + k - 1, TILE + k, 3 * TILE + 7,
            8 * TILE - 1, 8 * TILE, 8 * TILE + k - 1, 8 * TILE + k, 9 * TILE + 3, 16 * TILE + k, 20011,
            64 * TILE - 1, 64 * TILE, 64 * TILE + 1, 64 * TILE + k - 1, 70 * TILE + 5]
    for n in lens:
        b = rng.integers(0, 4, n).astype(np.uint8)
        if n >= 4 * k:
            b[n // 2:n // 2 + 2 * k] = 0                                  # a homopolymer run: hashF == hashR ties inside
        a, p, f = _iterate_arrays(sh, b)
        ek, ep, ef = oh.scan(b)
        assert np.array_equal(a, ek) and np.array_equal(p, ep) and np.array_equal(f, ef), (k, w, n)
    # many reads in a row through the same scratch (flag sequence numbers, output growth), alternating sizes
    for i in range(300):
        n = int(rng.integers(0, 3 * TILE)) if i % 7 else int(rng.integers(60 * TILE, 66 * TILE))
        b = rng.integers(0, 4, n).astype(np.uint8)
        a, p, f = _iterate_arrays(sh, b)
        ek, ep, ef = oh.scan(b)
        assert np.array_equal(a, ek) and np.array_equal(p, ep) and np.array_equal(f, ef), (k, w, n, i)


@pytest.mark.parametrize("k,w", [(21, 64), (31, 4), (19, 31)])
def test_iterator_latency_switch(k, w):
    """modRCiterator picks its leg by length (mg_host.c): reads below the crossover take the library's scalar loop, the
    others one kernel launch.  Lengths on both sides of several crossovers, both legs against the oracle and each other;
    FASTQ-style bytes above 3 go modulo 4 on both legs."""
    L = mg.lib()
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(k * 3 + w)
    assert DEFAULT_BELOW > 0
    for below in (DEFAULT_BELOW, 1, 151, 70000):
        for n in sorted({0, k - 1, k, 150, 151, below - 1, below, below + 1, 5000, 9000, 8191, 8192, 12287, 12288}):
            if n < 0:
                continue
            b = rng.integers(0, 4, n).astype(np.uint8)
            ek, ep, ef = oh.scan(b)
            for leg in (below, 0):
                a, p, f = _iterate_arrays(sh, b, leg)
                assert np.array_equal(a, ek) and np.array_equal(p, ep) and np.array_equal(f, ef), (k, w, n, below, leg)
    b = rng.integers(0, 4, 6000).astype(np.uint8)
    junk = b.copy(); junk[::5] |= 0xFC
    ref = _iterate_arrays(sh, b, 0)
    for leg in (0, 1 << 20):
        assert all(np.array_equal(x, y) for x, y in zip(ref, _iterate_arrays(sh, junk, leg)))
    assert L.mgIterHostBelow(-1) == DEFAULT_BELOW
