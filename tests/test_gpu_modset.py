"""GPU parity: the device modset (K3/K4/K5) and the C counterparts of the reference callers,
through the C ABI, vs the oracle and the golden outputs of the reference.  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import fasta, synth
from oracle import pyoracle as po
import util

pytestmark = pytest.mark.gpu


def oracle_build(oh, bits, batches):
    oms = po.Modset(oh, bits)
    tot = 0
    for bases, offs in batches:
        for r in range(len(offs) - 1):
            tot += oms.add_sequence(bases[offs[r]:offs[r + 1]])
    return oms, tot


def assert_same_modset(ms, oms, bits, check_index=True):
    L = mg.lib()
    mg.check(L.modsetSyncToHost(ms, 1 if check_index else 0))
    assert ms.contents.max == oms.max
    v, d, i = mg.modset_arrays(ms)
    assert np.array_equal(v[1:], oms.values()[1:]), "values / first-occurrence index order"
    assert np.array_equal(d[1:], oms.depths()[1:]), "depths"
    if check_index:
        idx = np.ctypeslib.as_array(ms.contents.index, (1 << bits,))
        assert np.array_equal(idx, oms.index_table()), "index[] slot layout (modset.c:51-57)"


def synth_batch(total, genome_bases, seed, err=0.03, n50=4000):
    genome = synth.iid_bases(genome_bases, seed)
    starts, offs, strands = synth.ont_read_plan(total, genome_bases, seed + 1, n50=n50, lo=30, hi=30000)
    return synth.reads_from_genome(genome, starts, offs, strands, err, seed + 2), offs.astype(np.int64)


@pytest.mark.parametrize("k,w,bits", [(21, 64, 22), (31, 4, 24), (19, 31, 22), (11, 1, 24), (16, 32, 20)])
def test_add_batches_vs_oracle(k, w, bits):
    """two successive batches into one modset (modutils -a f1 -a f2): indices continue at max+1"""
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b1 = synth_batch(600_000 if w > 2 else 150_000, 50_000, 11)
    b2 = synth_batch(300_000 if w > 2 else 80_000, 50_000, 11, err=0.05)     # same genome: many k-mers recur
    ms = mg.modsetCreate(sh, bits)
    n1 = mg.add_sequence_batch(ms, *b1)
    max1 = ms.contents.max
    n2 = mg.add_sequence_batch(ms, *b2)
    oms, tot = oracle_build(oh, bits, [b1, b2])
    assert n1 + n2 == tot and max1 <= ms.contents.max
    assert_same_modset(ms, oms, bits)
    mg.lib().modsetDestroy(ms)


def test_depth_saturation_and_duplicates_in_one_read():
    """a k-mer seen > 65535 times pins at 65535 (modutils.c:26); repeats inside one read keep the first index"""
    k, w, bits = 3, 1, 20
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    reads = [np.zeros(70000, np.uint8), np.array([0, 1, 2, 3] * 300, np.uint8), np.zeros(10, np.uint8)]
    batch = util.concat_reads(reads)
    ms = mg.modsetCreate(sh, bits)
    mg.add_sequence_batch(ms, *batch)
    oms, _ = oracle_build(oh, bits, [batch])
    assert oms.depths()[1] == 65535
    assert_same_modset(ms, oms, bits)
    # a second batch on top of a saturated depth stays saturated
    mg.add_sequence_batch(ms, *batch)
    oms2, _ = oracle_build(oh, bits, [batch, batch])
    assert_same_modset(ms, oms2, bits)


def test_find_batch_and_device_api():
    """modsetFindBatchDevice == modsetIndexFind(ms, kmer, false) incl. misses; AddBatch returns indices"""
    L = mg.lib()
    k, w, bits = 21, 16, 22
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    bases, offs = synth_batch(400_000, 40_000, 5)
    km, _, _, _ = util.oracle_scan_batch(oh, bases, offs)
    oms = po.Modset(oh, bits)
    expect = np.array([oms.find(x, True) for x in km], np.uint32)
    ms = mg.modsetCreate(sh, bits)
    d_k = mg.DeviceBuffer.from_numpy(km); d_i = mg.DeviceBuffer(len(km) * 4)
    mg.check(L.modsetAddBatchDevice(ms, d_k.ptr, len(km), d_i.ptr, 0, None))      # withDepth = 0: modmap.c:109
    assert np.array_equal(d_i.to_numpy(np.uint32, len(km)), expect)
    assert ms.contents.max == oms.max
    mg.check(L.modsetSyncToHost(ms, 0))
    assert not mg.modset_arrays(ms)[1].any()                                      # depth untouched
    rng = np.random.default_rng(0)
    probe = np.concatenate([km[::7], rng.integers(0, 1 << 42, 5000).astype(np.uint64)])
    want = np.array([oms.find(x) for x in probe], np.uint32)
    d_p = mg.DeviceBuffer.from_numpy(probe); d_o = mg.DeviceBuffer(len(probe) * 4)
    mg.check(L.modsetFindBatchDevice(ms, d_p.ptr, len(probe), d_o.ptr, None))
    got = d_o.to_numpy(np.uint32, len(probe))
    assert np.array_equal(got, want) and (got == 0).any() and (got != 0).any()
    # round trip: every stored value finds its own index
    vals = mg.modset_arrays(ms)[0][1:]
    d_v = mg.DeviceBuffer.from_numpy(vals); d_o2 = mg.DeviceBuffer(len(vals) * 4)
    mg.check(L.modsetFindBatchDevice(ms, d_v.ptr, len(vals), d_o2.ptr, None))
    assert np.array_equal(d_o2.to_numpy(np.uint32, len(vals)), np.arange(1, len(vals) + 1, dtype=np.uint32))


def test_host_scalar_and_device_batches_interleave():
    """scalar modsetIndexFind on the host arrays and GPU batches on the same Modset stay coherent"""
    L = mg.lib()
    k, w, bits = 19, 8, 20
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b1 = synth_batch(100_000, 20_000, 21)
    b2 = synth_batch(60_000, 20_000, 21, err=0.08)
    km2 = util.oracle_scan_batch(oh, *b2)[0]
    ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
    # host scalar inserts first (reference-style loop), then a GPU batch, then scalar again
    for x in km2[:500]:
        assert L.modsetIndexFind(ms, int(x), 1) == oms.find(x, True)
    mg.add_sequence_batch(ms, *b1)
    for r in range(len(b1[1]) - 1):
        oms.add_sequence(b1[0][b1[1][r]:b1[1][r + 1]])
    for x in km2[500:1500]:
        assert L.modsetIndexFind(ms, int(x), 1) == oms.find(x, True)
    for x in km2[:200]:
        assert L.modsetIndexFind(ms, int(x), 0) == oms.find(x, False)
    mg.add_sequence_batch(ms, *b2)
    for r in range(len(b2[1]) - 1):
        oms.add_sequence(b2[0][b2[1][r]:b2[1][r + 1]])
    assert_same_modset(ms, oms, bits)


def test_depth_is_current_after_a_batch_that_adds_no_entry():
    """the same reads added a second time leave max alone: the scalar lookup must still hand back an index whose
    ms->depth[] is current, because callers read and bump depth[index] themselves (modutils.c:26)"""
    L = mg.lib()
    k, w, bits = 21, 16, 20
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b1 = synth_batch(80_000, 20_000, 31)
    km = util.oracle_scan_batch(oh, *b1)[0]
    ms = mg.modsetCreate(sh, bits)
    mg.add_sequence_batch(ms, *b1)
    mg.check(L.modsetSyncToHost(ms, 1))                          # everything current: value[], depth[], index[]
    oms, _ = oracle_build(oh, bits, [b1, b1])
    max1 = ms.contents.max
    mg.add_sequence_batch(ms, *b1)                               # only re-hits: no new entry
    assert ms.contents.max == max1
    ix = L.modsetIndexFind(ms, int(km[0]), 0)                    # scalar API: must bring depth[] up to date
    assert ix == oms.find(km[0], False)
    assert int(ms.contents.depth[ix]) == int(oms.depths()[ix])
    assert_same_modset(ms, oms, bits)
    L.modsetDestroy(ms)


def test_histogram_clear_and_capacity(tmp_path):
    L = mg.lib()
    k, w, bits = 21, 4, 22
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b = synth_batch(500_000, 15_000, 31, err=0.01)
    ms = mg.modsetCreate(sh, bits)
    mg.add_sequence_batch(ms, *b)
    oms, _ = oracle_build(oh, bits, [b])
    d_h = mg.DeviceBuffer(65536 * 8)
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))          # before any sync: pending device counts
    assert np.array_equal(d_h.to_numpy(np.uint64, 65536), oms.histogram())
    tmp = str(tmp_path / "h.txt")
    with mg.CFile(tmp, "w") as f:
        L.mgDepthHistogram(ms, f)
    assert open(tmp).read() == oms.hist_text(str(tmp_path / "o.txt"))
    mg.check(L.modsetSyncToHost(ms, 0))
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))          # after a sync: same answer
    assert np.array_equal(d_h.to_numpy(np.uint64, 65536), oms.histogram())
    # clear = a fresh modsetCreate
    mg.check(L.mgModsetClear(ms, None))
    assert ms.contents.max == 0
    mg.add_sequence_batch(ms, *b)
    assert_same_modset(ms, oms, bits)
    # capacity: more distinct k-mers than size -> the reference's message (modset.c:58)
    small = mg.modsetCreate(mg.seqhashCreate(11, 1, 17), 20, 1000)
    rnd = np.random.default_rng(1).integers(0, 4, 20000).astype(np.uint8)
    with pytest.raises(mg.ModgpuError, match="hashTableSize 1000 is too small"):
        mg.add_sequence_batch(small, rnd, np.array([0, 20000], np.int64))


@pytest.mark.parametrize("tag", list(util.MODUTILS_TAGS))
def test_golden_modutils_flow(tag, golden_dir, tmp_path):
    """modutils -c .. -a reads.fa -a reads2.fa -wt -H -p 2 40 -H -wt through the C callers on the GPU"""
    L = mg.lib()
    B, k, w, s = util.MODUTILS_TAGS[tag]
    sh = mg.seqhashCreate(k, w, s)
    ms = mg.modsetCreate(sh, B)
    out = str(tmp_path / "out.txt")
    with mg.CFile(out, "w") as f:
        L.seqhashReport(sh, f)
        for fn in ("reads.fa", "reads2.fa"):
            names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, fn))
            assert L.mgAddSequences(ms, bases.ctypes.data, offs.ctypes.data, len(names), f) == 0
            L.modsetSummary(ms, f)
    tmp = str(tmp_path / "t.txt")

    def text(fn):
        with mg.CFile(tmp, "w") as f:
            fn(ms, f)
        return open(tmp).read()
    util.check_dump(text(L.mgModsetWriteText), "modutils_%s.dump.txt" % tag)
    assert text(L.mgDepthHistogram) == util.golden_text("modutils_%s.hist.txt" % tag)
    L.modsetDepthPrune(ms, 2, 40)
    with mg.CFile(out, "a") as f:
        L.modsetSummary(ms, f)
    assert text(L.mgDepthHistogram) == util.golden_text("modutils_%s.pruned_hist.txt" % tag)
    util.check_dump(text(L.mgModsetWriteText), "modutils_%s.pruned_dump.txt" % tag)
    assert open(out).read() == util.golden_text("modutils_%s.stdout.txt" % tag)
    # after the prune the device table is rebuilt from the host arrays on the next batch
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "reads2.fa"))
    before = ms.contents.max
    mg.add_sequence_batch(ms, bases, offs)
    assert ms.contents.max >= before


@pytest.mark.parametrize("tag", list(util.MODMAP_TAGS))
def test_golden_modmap_flow(tag, golden_dir, tmp_path):
    """modmap -K k -W w -S 17 -B 20 -f ref.fa -q queries.fa: same report, Q and M lines as the reference"""
    L = mg.lib()
    k, w = util.MODMAP_TAGS[tag]
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, 20)
    ref = L.mgReferenceCreate(ms, 1 << 26)
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "ref.fa"))
    qn, qb, qo = fasta.read_fasta(os.path.join(golden_dir, "queries.fa"))
    out = str(tmp_path / "mm.txt")
    with mg.CFile(out, "w") as f:
        _libc = C.CDLL(None)
        _libc.fprintf.argtypes = [C.c_void_p, C.c_char_p]
        _libc.fprintf(f, ("  modmap initialised with k = %d, w = %d, random seed = 17\n" % (k, w)).encode())
        cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
        assert L.mgReferenceRead(ref, bases.ctypes.data, offs.ctypes.data, len(names), cn, True, f) == 0
        cq = (C.c_char_p * len(qn))(*[n.encode() for n in qn])
        assert L.mgQueryProcess(ref, qb.ctypes.data, qo.ctypes.data, len(qn), cq, f) == 0
    assert open(out).read() == util.golden_text("modmap_%s.stdout.txt" % tag)
    L.mgReferenceDestroy(ref)


@pytest.mark.parametrize("k,w,bits,path", [(21, 32, 22, "direct"), (21, 32, 22, "part"), (19, 31, 23, "part"), (15, 8, 24, "part"),
                                            (27, 4, 22, "part"), (31, 4, 22, "part"),
                                            (21, 32, 22, "2 levels"), (19, 31, 23, "2 levels"), (15, 8, 24, "2 levels"), (27, 4, 22, "2 levels"), (21, 64, 28, "2 levels"),
                                            # round 6: the two-level lookups read an 8-byte copy of the table where 2k - log2 NB <= 32 (16 buckets here: k <= 18): the widest key that fits, and one more bit
                                            (18, 8, 22, "2 levels"), (17, 4, 22, "2 levels"), (16, 8, 22, "2 levels")])
def test_seed_lists_vs_oracle(k, w, bits, path):
    """mgQueryReadsDevice: Seed{index,pos} per read incl. misses (modmap.c:197-206) -- by direct probes in ordinal order and by
    the partitioned lookup (mgTableFindPartitioned: first partition pass of the build on the query's modimizers, lookups bin by
    bin against one piece of the table, results pulled back into ordinal order through the scatter's run table); k = 31 has no
    one-word element and must fall back to the direct probes by itself"""
    L = mg.lib()
    with mg.knobs(FIND_PATH=path):
        _seed_lists_vs_oracle(L, k, w, bits)


def _seed_lists_vs_oracle(L, k, w, bits):
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    refb = synth_batch(300_000, 300_000, 41, err=0.0, n50=50_000)
    ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
    for r in range(len(refb[1]) - 1):
        for x in oh.scan(refb[0][refb[1][r]:refb[1][r + 1]])[0]:
            oms.find(x, True)
    km_ref = util.oracle_scan_batch(oh, *refb)[0]
    d_k = mg.DeviceBuffer.from_numpy(km_ref)
    mg.check(L.modsetAddBatchDevice(ms, d_k.ptr, len(km_ref), None, 0, None))
    q = synth_batch(200_000, 300_000, 41, err=0.04, n50=6000)       # reads of the same genome, with errors
    qk, qp, _, qst = util.oracle_scan_batch(oh, *q)
    want = np.array([oms.find(x) for x in qk], np.uint32)
    total = int(q[1][-1])
    d_p = mg.DeviceBuffer.from_numpy(mg.pack_host(q[0])); d_o = mg.DeviceBuffer.from_numpy(q[1].astype(np.uint64))
    cap = len(qk) + 5
    d_ix = mg.DeviceBuffer(cap * 4); d_pos = mg.DeviceBuffer(cap * 4); d_rid = mg.DeviceBuffer(cap * 4)
    n = C.c_uint64()
    mg.check(L.mgQueryReadsDevice(ms, d_p.ptr, total, d_o.ptr, len(q[1]) - 1, d_ix.ptr, d_pos.ptr, d_rid.ptr, cap, C.byref(n), None))
    assert n.value == len(qk)
    assert np.array_equal(d_ix.to_numpy(np.uint32, n.value), want)
    assert np.array_equal(d_pos.to_numpy(np.uint32, n.value) & mg.MG_POS_MASK, qp.astype(np.uint32))
    rid = d_rid.to_numpy(np.uint32, n.value)
    assert np.array_equal(np.searchsorted(rid, np.arange(len(q[1]))), qst)
    assert (want == 0).any() and (want != 0).any()
    # a second, larger query batch against the same table (other sub-chunk counts, a last partial sub-chunk), twice
    q = synth_batch(1_500_000, 300_000, 41, err=0.02, n50=3000)
    qk = util.oracle_scan_batch(oh, *q)[0]
    order = np.argsort(oms.values()[1:], kind="stable"); vs = oms.values()[1:][order]
    at = np.minimum(np.searchsorted(vs, qk), len(vs) - 1)
    want = np.where(vs[at] == qk, order[at] + 1, 0).astype(np.uint32)
    total = int(q[1][-1])
    d_p = mg.DeviceBuffer.from_numpy(mg.pack_host(q[0])); d_o = mg.DeviceBuffer.from_numpy(q[1].astype(np.uint64))
    cap = len(qk) + 5
    d_ix = mg.DeviceBuffer(cap * 4)
    for rep in range(2):
        mg.check(L.mgQueryReadsDevice(ms, d_p.ptr, total, d_o.ptr, len(q[1]) - 1, d_ix.ptr, None, None, cap, C.byref(n), None))
        assert n.value == len(qk) and np.array_equal(d_ix.to_numpy(np.uint32, n.value), want), rep


@pytest.mark.parametrize("path", ["direct", "2 levels", None])
def test_query_batches_pipelined_equal_synchronous(path):
    """mgQueryReadsDeviceAsync / Wait (the scan of batch i + 1 beside the lookups of batch i, two scratch arenas): five batches of
    different sizes -- one of them empty, one whose survivor guess is too small (d = 4 reads full of one k-mer would do; here a
    capacity the caller under-sizes is the error case) -- give, seed for seed, what mgQueryReadsDevice gives; while a ticket is out
    the modset refuses other batch calls; tickets are waited for in order."""
    L = mg.lib()
    k, w, bits = 21, 32, 22
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, bits)
    refb = synth_batch(300_000, 300_000, 41, err=0.0, n50=50_000)
    d_rp = mg.DeviceBuffer.from_numpy(mg.pack_host(refb[0])); d_ro = mg.DeviceBuffer.from_numpy(refb[1].astype(np.uint64))
    nh = C.c_uint64()
    mg.check(L.mgAddReadsDevice(ms, d_rp.ptr, int(refb[1][-1]), d_ro.ptr, len(refb[1]) - 1, C.byref(nh), None))
    with mg.knobs(FIND_PATH=path):
        batches = []
        for i, (nb, n50, err) in enumerate([(400_000, 6000, 0.03), (1_200_000, 3000, 0.02), (0, 1, 0), (150_000, 800, 0.05), (900_000, 20_000, 0.01)]):
            if nb:
                q = synth_batch(nb, 300_000, 41, err=err, n50=n50)
                d_p = mg.DeviceBuffer.from_numpy(mg.pack_host(q[0])); d_o = mg.DeviceBuffer.from_numpy(q[1].astype(np.uint64))
                total, nr = int(q[1][-1]), len(q[1]) - 1
            else:
                d_p = mg.DeviceBuffer(64); d_o = mg.DeviceBuffer.from_numpy(np.zeros(1, np.uint64)); total, nr = 0, 0
            cap = total // w * 2 + 1000
            outs = [mg.DeviceBuffer(cap * 4) for _ in range(3)]
            n = C.c_uint64()
            mg.check(L.mgQueryReadsDevice(ms, d_p.ptr, total, d_o.ptr, nr, outs[0].ptr, outs[1].ptr, outs[2].ptr, cap, C.byref(n), None))
            want = [o.to_numpy(np.uint32, n.value) for o in outs]
            batches.append((d_p, d_o, total, nr, cap, want))
        tickets, outs_all = [], []

        def start(i):
            d_p, d_o, total, nr, cap, _ = batches[i]
            outs = [mg.DeviceBuffer(cap * 4) for _ in range(3)]
            t = C.c_void_p()
            mg.check(L.mgQueryReadsDeviceAsync(ms, d_p.ptr, total, d_o.ptr, nr, outs[0].ptr, outs[1].ptr, outs[2].ptr, cap, C.byref(t), None))
            tickets.append(t); outs_all.append(outs)
        start(0)
        for i in range(len(batches)):
            if i + 1 < len(batches):
                start(i + 1)
                if i == 0:                                    # two in flight: a third is refused, and so is any other batch call
                    t3 = C.c_void_p()
                    assert L.mgQueryReadsDeviceAsync(ms, batches[0][0].ptr, batches[0][2], batches[0][1].ptr, batches[0][3], outs_all[0][0].ptr,
                                                     None, None, batches[0][4], C.byref(t3), None) != 0
                    assert L.mgQueryReadsDevice(ms, batches[0][0].ptr, batches[0][2], batches[0][1].ptr, batches[0][3], outs_all[0][0].ptr,
                                                None, None, batches[0][4], C.byref(nh), None) != 0 and b"in flight" in L.mgLastError()
            n = C.c_uint64()
            mg.check(L.mgQueryReadsDeviceWait(tickets[i], C.byref(n), None))
            want = batches[i][5]
            assert n.value == len(want[0]), i
            for o, w_ in zip(outs_all[i], want):
                assert np.array_equal(o.to_numpy(np.uint32, n.value), w_), i
        # a capacity too small for the batch's seeds: the error of the synchronous call, and the modset is free again afterwards
        d_p, d_o, total, nr, cap, want = batches[1]
        t = C.c_void_p(); o = mg.DeviceBuffer(400)
        mg.check(L.mgQueryReadsDeviceAsync(ms, d_p.ptr, total, d_o.ptr, nr, o.ptr, None, None, 100, C.byref(t), None))
        n = C.c_uint64()
        assert L.mgQueryReadsDeviceWait(t, C.byref(n), None) != 0 and n.value == len(want[0])
        o3 = [mg.DeviceBuffer(cap * 4) for _ in range(3)]
        mg.check(L.mgQueryReadsDevice(ms, d_p.ptr, total, d_o.ptr, nr, o3[0].ptr, o3[1].ptr, o3[2].ptr, cap, C.byref(n), None))
        assert np.array_equal(o3[0].to_numpy(np.uint32, n.value), want[0])
    L.modsetDestroy(ms)


def test_full_size_build_properties():
    """BASELINE config 2 at full size (10 Gbp, table bits 30) plus a 1 Gbp run: properties that
    hold for any correct modset build — sum of depths == number of modimizers (no saturation here),
    every stored k-mer finds its own index, re-adding the same batch adds no entries and doubles
    the depth sum, the DP histogram sums to max and its weighted sum to the depth sum."""
    L = mg.lib()
    sh = mg.seqhashCreate(21, 64, 17)
    for gbp, bits in ((1.0, 28), (10.0, 30)):
        total = int(gbp * 1e9); G = total // 30
        starts, offs, strands = synth.ont_read_plan(total, G, 1000)
        d_g = mg.DeviceBuffer(L.mgPackedWords(G) * 4)
        mg.check(L.mgSynthGenome(d_g.ptr, G, 12345, None))
        d_s = mg.DeviceBuffer.from_numpy(starts); d_of = mg.DeviceBuffer.from_numpy(offs); d_st = mg.DeviceBuffer.from_numpy(strands)
        d_r = mg.DeviceBuffer(L.mgPackedWords(total) * 4)
        mg.check(L.mgSynthReads(d_g.ptr, G, d_s.ptr, d_of.ptr, d_st.ptr, len(starts), total, 0.05, 777, d_r.ptr, None))
        d_g.free()
        ms = mg.modsetCreate(sh, bits)
        n = C.c_uint64()
        mg.check(L.mgAddReadsDevice(ms, d_r.ptr, total, d_of.ptr, len(starts), C.byref(n), None))
        S, U = n.value, ms.contents.max
        assert abs(S / (total / 64.0) - 1) < 0.01 and 0 < U <= S
        d_h = mg.DeviceBuffer(65536 * 8)
        mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
        mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
        h = d_h.to_numpy(np.uint64, 65536)
        assert int(h.sum()) == U and h[0] == 0
        assert int((h * np.arange(65536, dtype=np.uint64)).sum()) == S or h[65535] > 0
        mg.check(L.mgAddReadsDevice(ms, d_r.ptr, total, d_of.ptr, len(starts), C.byref(n), None))
        assert n.value == S and ms.contents.max == U                      # idempotent key set
        mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
        mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
        h2 = d_h.to_numpy(np.uint64, 65536)
        assert int(h2.sum()) == U and (h2[1::2][:30000].sum() == 0 or h2[65535] > 0)      # every depth doubled -> even
        if gbp == 1.0:
            mg.check(L.modsetSyncToHost(ms, 0))
            v, d, _ = mg.modset_arrays(ms)
            assert len(np.unique(v[1:])) == U and int(d[1:].astype(np.int64).sum()) == 2 * S
            d_v = mg.DeviceBuffer.from_numpy(v[1:]); d_o = mg.DeviceBuffer(U * 4)
            mg.check(L.modsetFindBatchDevice(ms, d_v.ptr, U, d_o.ptr, None))
            assert np.array_equal(d_o.to_numpy(np.uint32, U), np.arange(1, U + 1, dtype=np.uint32))
        L.modsetDestroy(ms)
        d_r.free()


def test_table_growth_and_lazy_zeroing():
    """the device table starts small and grows by rehashing; buckets nothing wrote to are zeroed
    lazily.  Interleave large (bucketed) and tiny (atomic path) batches, lookups and a clear."""
    L = mg.lib()
    k, w, bits = 15, 1, 24                      # every k-mer start is a modimizer: many distinct keys quickly
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(123)
    ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
    sizes = [300, 40_000, 50, 200_000, 7, 600_000, 1000]      # crosses the 2^16-slot start size several times
    for i, n in enumerate(sizes):
        b = rng.integers(0, 4, n).astype(np.uint8)
        offs = np.array([0, n], np.int64)
        assert mg.add_sequence_batch(ms, b, offs) == oms.add_sequence(b)
        assert ms.contents.max == oms.max
        probe = np.concatenate([oh.scan(b)[0][:500], rng.integers(0, 1 << 30, 200).astype(np.uint64)])
        d_p = mg.DeviceBuffer.from_numpy(probe); d_o = mg.DeviceBuffer(len(probe) * 4)
        mg.check(L.modsetFindBatchDevice(ms, d_p.ptr, len(probe), d_o.ptr, None))
        assert np.array_equal(d_o.to_numpy(np.uint32, len(probe)), np.array([oms.find(x) for x in probe], np.uint32)), i
    assert_same_modset(ms, oms, bits)
    # clear, then a tiny batch first (atomic path on a table whose buckets are all "never written")
    mg.check(L.mgModsetClear(ms, None))
    oms2 = po.Modset(oh, bits)
    for n in (40, 90_000, 13):
        b = rng.integers(0, 4, n).astype(np.uint8)
        assert mg.add_sequence_batch(ms, b, np.array([0, n], np.int64)) == oms2.add_sequence(b)
    assert_same_modset(ms, oms2, bits)


def test_chunked_insert_matches(monkeypatch):
    """inserts are split into passes of at most MG_ADD_CHUNK modimizers; a tiny chunk size must not change anything"""
    import subprocess, sys, os
    code = r"""
import numpy as np, modimizer_amd as mg
from oracle import pyoracle as po
from modimizer_amd import synth
sh = mg.seqhashCreate(17, 4, 17); oh = po.Hasher(17, 4, 17)
g = synth.iid_bases(30000, 3)
st, offs, sd = synth.ont_read_plan(400000, len(g), 4, n50=3000, lo=50, hi=9000)
b = synth.reads_from_genome(g, st, offs, sd, 0.02, 5)
ms = mg.modsetCreate(sh, 22); oms = po.Modset(oh, 22)
n = mg.add_sequence_batch(ms, b, offs.astype(np.int64))
t = sum(oms.add_sequence(b[int(offs[r]):int(offs[r+1])]) for r in range(len(st)))
mg.check(mg.lib().modsetSyncToHost(ms, 1))
v, d, _ = mg.modset_arrays(ms)
assert n == t and ms.contents.max == oms.max
assert np.array_equal(v[1:], oms.values()[1:]) and np.array_equal(d[1:], oms.depths()[1:])
assert np.array_equal(np.ctypeslib.as_array(ms.contents.index, (1 << 22,)), oms.index_table())
print("chunked ok", n)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for chunk, path in (("7000", "bucket"), ("5000", "direct"), ("33333", "auto")):
        env = dict(os.environ, MODGPU_ADD_CHUNK=chunk, MODGPU_TABLE_PATH=path, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "chunked ok" in r.stdout, (chunk, path, r.stderr[-1500:])


def test_table_regrown_before_lookups():
    """mgAddReadsDevice sizes the device table for occurrences / 0.75 (a set being built and counted); a lookup batch first
    brings it to entries / 0.4 (rehash on the device).  Since round 6 the slot count is NB x R with R any multiple of 64, not the
    next power of two: 90 000 distinct k-mers take 120 832 slots after the build (load 0.74), 225 000 or so after the first lookup;
    indices, values and depths unchanged, found and absent k-mers answered as before"""
    with mg.knobs(TABLE_LOAD=None, TIGHT_LOAD=None, TABLE_PATH=None, BUCKET_R=None):      # (the slot counts asserted are the defaults': tools/test_paths.sh runs the suite under sizing knobs)
        _table_regrown_before_lookups()


def _table_regrown_before_lookups():
    L = mg.lib()
    k, w, bits = 21, 4, 22
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    rng = np.random.default_rng(123)
    b = rng.integers(0, 4, 360_000).astype(np.uint8)
    offs = np.array([0, len(b)], np.int64)
    ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
    assert mg.add_sequence_batch(ms, b, offs) == oms.add_sequence(b)
    slots0 = L.mgModsetDeviceSlots(ms)
    assert oms.max / 0.75 <= slots0 < oms.max / 0.75 * 1.04 and slots0 % 64 == 0 and slots0 & (slots0 - 1), (slots0, oms.max)
    present = oms.values()[1:20001].copy()
    absent = (present ^ np.uint64(0x15555)) & np.uint64((1 << (2 * k)) - 1)
    q = np.concatenate([present, absent])
    d_q = mg.DeviceBuffer.from_numpy(q); d_o = mg.DeviceBuffer(len(q) * 4)
    mg.check(L.modsetFindBatchDevice(ms, d_q.ptr, len(q), d_o.ptr, None))
    got = d_o.to_numpy(np.uint32, len(q))
    want = np.array([oms.find(int(x)) for x in q], np.uint32)
    assert np.array_equal(got, want)
    slots1 = L.mgModsetDeviceSlots(ms)
    assert oms.max / 0.4 <= slots1 < oms.max / 0.4 * 1.04 and slots1 > slots0, (slots1, oms.max)
    b2 = rng.integers(0, 4, 50_000).astype(np.uint8)
    assert mg.add_sequence_batch(ms, b2, np.array([0, len(b2)], np.int64)) == oms.add_sequence(b2)
    assert_same_modset(ms, oms, bits)


def test_rank_slices():
    """bucketed path: the rank lookups run per slice of the ordinal range (mg_table.hip, mgRankLookupKernel).  One slice
    (a bucket's whole list in one group, far longer than a wave), the most slices the kernel takes (63 + the group of
    k-mers already in the table), and something in between -- two batches each, the second mostly k-mers of the first"""
    import subprocess, sys, os
    code = r"""
import numpy as np, modimizer_amd as mg
from oracle import pyoracle as po
from modimizer_amd import synth
sh = mg.seqhashCreate(17, 4, 17); oh = po.Hasher(17, 4, 17)
g = synth.iid_bases(30000, 3)
st, offs, sd = synth.ont_read_plan(400000, len(g), 4, n50=3000, lo=50, hi=9000)
b = synth.reads_from_genome(g, st, offs, sd, 0.02, 5)
ms = mg.modsetCreate(sh, 22); oms = po.Modset(oh, 22)
half = len(st) // 2
n = mg.add_sequence_batch(ms, b[:int(offs[half])], offs[:half + 1].astype(np.int64))
n += mg.add_sequence_batch(ms, b, offs.astype(np.int64))
t = sum(oms.add_sequence(b[int(offs[r]):int(offs[r+1])]) for r in range(half))
t += sum(oms.add_sequence(b[int(offs[r]):int(offs[r+1])]) for r in range(len(st)))
mg.check(mg.lib().modsetSyncToHost(ms, 1))
v, d, _ = mg.modset_arrays(ms)
assert n == t and ms.contents.max == oms.max
assert np.array_equal(v[1:], oms.values()[1:]) and np.array_equal(d[1:], oms.depths()[1:])
assert np.array_equal(np.ctypeslib.as_array(ms.contents.index, (1 << 22,)), oms.index_table())
print("slices ok", n)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for shift in ("30", "1", "11"):
        env = dict(os.environ, MODGPU_RANK_SLICE_SHIFT=shift, MODGPU_TABLE_PATH="bucket", PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "slices ok" in r.stdout, (shift, r.stderr[-1500:])
    # buckets of 8192 slots -- the geometry a table of 2^31 slots takes (table bits 32 with more than 6.4e8 entries: at most
    # 2^18 buckets) -- forced on this small table: the dedup kernel's threads then hold eight slots each (its second instance),
    # in both flag polarities and with the merge kernel taking the dedup kernel's slots; and smaller buckets (256-thread shape)
    for knobs in ({"MODGPU_BUCKET_R": "8192"}, {"MODGPU_BUCKET_R": "8192", "MODGPU_FLAG_POLARITY": "1", "MODGPU_MERGE_SLOTS": "1"},
                  {"MODGPU_BUCKET_R": "8192", "MODGPU_PART_PACKED": "0"}, {"MODGPU_BUCKET_R": "1024", "MODGPU_BUCKET_T": "256"},
                  {"MODGPU_PART_BIG": "0"}):                     # the partition passes with sub-chunks of 8192 elements where they would take 16384
        env = dict(os.environ, MODGPU_TABLE_PATH="bucket", PYTHONPATH=root, **knobs)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "slices ok" in r.stdout, (knobs, r.stderr[-1500:])


def test_large_table_bits_geometry():
    """table bits 31/32 (the device limit): the device table is sized by content, so this is cheap"""
    L = mg.lib()
    sh = mg.seqhashCreate(21, 16, 17); oh = po.Hasher(21, 16, 17)
    b = synth_batch(400_000, 30_000, 77)
    for bits in (31, 32):
        ms = mg.modsetCreate(sh, bits)
        n = mg.add_sequence_batch(ms, *b)
        oms = po.Modset(oh, 24)
        t = sum(oms.add_sequence(b[0][b[1][r]:b[1][r + 1]]) for r in range(len(b[1]) - 1))
        assert n == t and ms.contents.max == oms.max
        mg.check(L.modsetSyncToHost(ms, 0))
        v, d, _ = mg.modset_arrays(ms)
        assert np.array_equal(v[1:], oms.values()[1:]) and np.array_equal(d[1:], oms.depths()[1:])
        L.modsetDestroy(ms)
    with pytest.raises(mg.ModgpuError, match="table bits 20..32"):
        ms = mg.modsetCreate(sh, 33)
        mg.add_sequence_batch(ms, *b)


def test_merge_and_prune_on_device_vs_golden(golden_dir, tmp_path):
    """modsetMerge / modsetDepthPrune / modsetPack with the sets built and living on the GPU: same
    values, depths, info bits, index[] layout and summaries as the reference (tests/golden/modset_ops.npz)"""
    L = mg.lib()
    g = np.load(os.path.join(util.GOLDEN, "modset_ops.npz"))
    k, w, seed, B = (int(x) for x in g["params"])
    sh = mg.seqhashCreate(k, w, seed)
    a, b = mg.modsetCreate(sh, B), mg.modsetCreate(sh, B)
    for ms, fn in ((a, "reads.fa"), (b, "reads2.fa")):
        names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, fn))
        mg.add_sequence_batch(ms, bases, offs)
        mg.check(L.modsetSyncToHost(ms, 0))
    for i in range(1, b.contents.max + 1):
        b.contents.info[i] = (i % 4) | ((i % 3 == 0) * 8)
    for i in range(1, a.contents.max + 1):
        a.contents.info[i] = ((i // 2) % 4) | ((i % 5 == 0) * 16)
    tmp = str(tmp_path / "s.txt")

    def check(ms, tag):
        mg.check(L.modsetSyncToHost(ms, 1))
        v, d, i = mg.modset_arrays(ms)
        assert np.array_equal(v[1:], g[tag + "_value"][1:]), tag
        assert np.array_equal(d, g[tag + "_depth"]) and np.array_equal(i, g[tag + "_info"]), tag
        idx = np.ctypeslib.as_array(ms.contents.index, (1 << B,))
        nz = np.nonzero(idx)[0]
        assert np.array_equal(nz.astype(np.uint32), g[tag + "_index_pos"]) and np.array_equal(idx[nz], g[tag + "_index_val"]), tag
        with mg.CFile(tmp, "w") as f:
            L.modsetSummary(ms, f)
        assert open(tmp, "rb").read() == g[tag + "_summary"].tobytes(), tag
    check(a, "a"); check(b, "b")
    assert L.modsetMerge(a, b)                      # a lives on the device: merged there
    check(a, "merged")
    L.modsetDepthPrune(a, 2, 30)                    # compacted and re-tabled on the device
    check(a, "pruned")
    assert L.modsetPack(a) and a.contents.size == int(g["packed_size"][0])
    # and the device table still answers lookups after all that
    vals = mg.modset_arrays(a)[0][1:]
    d_v = mg.DeviceBuffer.from_numpy(vals); d_o = mg.DeviceBuffer(len(vals) * 4)
    mg.check(L.modsetFindBatchDevice(a, d_v.ptr, len(vals), d_o.ptr, None))
    assert np.array_equal(d_o.to_numpy(np.uint32, len(vals)), np.arange(1, len(vals) + 1, dtype=np.uint32))


def test_rank_order_merge_from_device_arrays():
    """what the root of mgModsetMergeRankOrder does with a peer's arrays (per-GPU sets over contiguous blocks of reads,
    merged in rank order = the single-stream build): the second block's set handed over as DEVICE arrays (mgModsetMergeDeviceArrays)
    gives value[] / depth[] / info bits / index[] of the oracle's build over both blocks, and the same as the host-array form."""
    L = mg.lib()
    k, w, bits = 21, 64, 22
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b1 = synth_batch(900_000, 60_000, 21)
    b2 = synth_batch(700_000, 60_000, 21, err=0.05)            # same genome: most k-mers recur, some are new
    sets = []
    for form in ("device", "host"):
        a, b = mg.modsetCreate(sh, bits), mg.modsetCreate(sh, bits)
        mg.add_sequence_batch(a, *b1); mg.add_sequence_batch(b, *b2)
        mg.check(L.modsetSyncToHost(b, 0))
        n2 = b.contents.max
        bv, bd, bi = mg.modset_arrays(b)
        bi[1:] = (np.arange(1, n2 + 1) % 4).astype(np.uint8)     # copy bits travel too (modset.c:123-126)
        mg.check(L.modsetSyncToHost(a, 0))
        ai = mg.modset_arrays(a)[2]; ai[1:] = ((np.arange(1, a.contents.max + 1) // 3) % 4).astype(np.uint8)
        if form == "device":
            dv, dd, di = (mg.DeviceBuffer.from_numpy(x[1:].copy()) for x in (bv, bd, bi))
            assert L.mgModsetMergeDeviceArrays(a, dv.ptr, dd.ptr, di.ptr, n2)
        else:
            assert L.mgModsetMergeArrays(a, bv.ctypes.data, bd.ctypes.data, bi.ctypes.data, n2)
        mg.check(L.modsetSyncToHost(a, 1))
        v, d, i = (x.copy() for x in mg.modset_arrays(a))
        idx = np.ctypeslib.as_array(a.contents.index, (1 << bits,)).copy()
        sets.append((a.contents.max, v, d, i, idx))
        if form == "device":
            oms, _ = oracle_build(oh, bits, [b1, b2])
            assert_same_modset(a, oms, bits)
        L.modsetDestroy(a); L.modsetDestroy(b)
    (m0, v0, d0, i0, x0), (m1, v1, d1, i1, x1) = sets
    assert m0 == m1 and np.array_equal(v0[1:m0 + 1], v1[1:m0 + 1]) and np.array_equal(d0[:m0 + 1], d1[:m0 + 1])
    assert np.array_equal(i0[:m0 + 1], i1[:m0 + 1]) and np.array_equal(x0, x1)
    assert i0[1:m0 + 1].max() == 3 and (i0[1:m0 + 1] & ~np.uint8(3)).max() == 0
    # a set that lives on the host alone: the device form declines, nothing changes
    c = mg.modsetCreate(sh, bits)
    dv = mg.DeviceBuffer.from_numpy(np.arange(4, dtype=np.uint64))
    assert not L.mgModsetMergeDeviceArrays(c, dv.ptr, dv.ptr, dv.ptr, 2) and c.contents.max == 0
    L.modsetDestroy(c)


@pytest.mark.gpu
def test_sender_of_a_rank_order_merge_keeps_its_host_depths():
    """what a non-root rank of mgModsetMergeRankOrder does to its own set (mg_comm.hip -> mgHookDeviceView: the pending device
    counts folded so that value[] / depth[] can leave from the device table) must leave the caller's depth[] what a sync would have made
    it -- "the other ranks' sets are left as they are" -- although the fold clears the pending flag.  Counts added afterwards still land."""
    L = mg.lib()
    k, w, bits = 21, 64, 22
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    b1 = synth_batch(900_000, 60_000, 21)
    b2 = synth_batch(300_000, 60_000, 21, err=0.05)
    ms = mg.modsetCreate(sh, bits)
    mg.add_sequence_batch(ms, *b1)                                # counted on the device, never synced: the usual state before a merge
    dv, dd, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
    assert L.mgHookDeviceView(ms, C.byref(dv), C.byref(dd), C.byref(n)) == 0 and n.value == ms.contents.max
    oms, _ = oracle_build(oh, bits, [b1])
    dev_depth = np.empty(n.value, np.uint16)
    mg.check(L.mgMemcpyD2H(dev_depth.ctypes.data, dd, dev_depth.nbytes, None))
    assert np.array_equal(dev_depth, oms.depths()[1:])           # what the sender ships
    mg.check(L.modsetSyncToHost(ms, 1))
    assert_same_modset(ms, oms, bits)                             # ... and what it keeps
    mg.add_sequence_batch(ms, *b2)
    oms2, _ = oracle_build(oh, bits, [b1, b2])
    mg.check(L.modsetSyncToHost(ms, 1))
    assert_same_modset(ms, oms2, bits)
    L.modsetDestroy(ms)


@pytest.mark.gpu
def test_modmap_query_host_chain_path(golden_dir, tmp_path):
    """the path taken when a read has more blocks than the device chaining keeps (seed lists chained on the
    host): same lines.  Forced through MODGPU_QUERY_HOST_CHAIN=1 in a fresh process."""
    import subprocess, sys
    out = str(tmp_path / "mm.txt")
    code = (
        "import ctypes as C, os, modimizer_amd as mg\n"
        "from modimizer_amd import fasta\n"
        "L = mg.lib(); sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 20)\n"
        "ref = L.mgReferenceCreate(ms, 1 << 26)\n"
        "with mg.CFile(%r, 'w') as f:\n"
        "    assert L.mgReferenceFastaRead(ref, %r, True, f) == 0\n"
        "    assert L.mgQueryFile(ref, %r, f) == 0\n"
    ) % (out, os.path.join(golden_dir, "ref.fa").encode(), os.path.join(golden_dir, "queries.fa").encode())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                       env=dict(os.environ, MODGPU_QUERY_HOST_CHAIN="1"))
    assert r.returncode == 0, r.stderr[-800:]
    assert open(out).read().splitlines() == util.golden_text("modmap_k21d64.stdout.txt").splitlines()[1:]


@pytest.mark.gpu
def test_modmap_query_many_blocks_overflow(tmp_path):
    """a read stitched from 40 short pieces of the reference in shuffled order: more M blocks than the device
    kernel keeps per read, so the batch goes the long way; both ways must print the same lines"""
    import subprocess, sys
    rng = np.random.default_rng(12)
    g = rng.integers(0, 4, 400_000).astype(np.uint8)
    pieces = [g[a:a + 3000] for a in rng.permutation(np.arange(0, 390_000, 9000))[:40]]
    reads = [np.concatenate(pieces), g[1000:9000], np.concatenate(pieces[:5])]
    fa_ref, fa_q = str(tmp_path / "ref.fa"), str(tmp_path / "q.fa")
    fasta.write_fasta(fa_ref, ["chr"], [g]); fasta.write_fasta(fa_q, ["stitched", "plain", "five"], reads)
    outs = []
    for knob in ("0", "1"):
        out = str(tmp_path / ("o%s.txt" % knob))
        code = (
            "import modimizer_amd as mg\n"
            "L = mg.lib(); sh = mg.seqhashCreate(15, 8, 17); ms = mg.modsetCreate(sh, 20)\n"
            "ref = L.mgReferenceCreate(ms, 1 << 26)\n"
            "with mg.CFile(%r, 'w') as f:\n"
            "    assert L.mgReferenceFastaRead(ref, %r, True, f) == 0\n"
            "    assert L.mgQueryFile(ref, %r, f) == 0\n"
        ) % (out, fa_ref.encode(), fa_q.encode())
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                           env=dict(os.environ, MODGPU_QUERY_HOST_CHAIN=knob))
        assert r.returncode == 0, r.stderr[-800:]
        outs.append(open(out).read())
    assert outs[0] == outs[1]
    assert sum(l.startswith("M\tstitched") for l in outs[0].splitlines()) > 16


@pytest.mark.gpu
def test_histogram_kept_by_the_build(tmp_path):
    """a set built by one add from empty keeps its depth histogram while it is built (merge kernel): depths 1, 2,
    mid-range, above the LDS bins, and saturated at 65535; a second add falls back to the table pass"""
    L = mg.lib()
    k, w, bits = 15, 1, 22
    rng = np.random.default_rng(21)
    g = rng.integers(0, 4, 40_000).astype(np.uint8)
    reads = [np.zeros(70_000, np.uint8)]                                   # one k-mer 69 986 times -> 65535
    reads += [g[:30_000]] * 1 + [g[:20_000]] * 2 + [g[:5_000]] * 300       # depths 1, 3, 303
    reads += [np.tile(g[100:140], 12)]                                     # short repeats
    bases, offs = util.concat_reads(reads)
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    oms = po.Modset(oh, bits)
    for r in reads:
        oms.add_sequence(r)
    ms = mg.modsetCreate(sh, bits)
    mg.add_sequence_batch(ms, bases, offs)
    d_h = mg.DeviceBuffer(65536 * 8)
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
    want = oms.histogram()
    assert want[65535] >= 1 and want[303:].sum() > 10 and want[1] > 0
    assert np.array_equal(d_h.to_numpy(np.uint64, 65536), want)
    # a second add: the table pass takes over
    mg.add_sequence_batch(ms, bases[:30_000], np.array([0, 30_000], np.int64))
    oms.add_sequence(bases[:30_000])
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
    assert np.array_equal(d_h.to_numpy(np.uint64, 65536), oms.histogram())
    # cleared and rebuilt: kept again
    mg.check(L.mgModsetClear(ms, None))
    mg.add_sequence_batch(ms, bases, offs)
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
    assert np.array_equal(d_h.to_numpy(np.uint64, 65536), want)
    L.modsetDestroy(ms); oms.close()


def test_kmers_with_very_many_copies():
    """bucketed path (and, last, the atomic one), the dedup kernel's loop over what was not fetched ahead (mg_table.hip, mgDedupCountWave / MG_HOT_DEPTH): buckets with
    far more occurrences than a workgroup fetches ahead (3 per thread) -- thousands of copies of one read (whole waves of one k-mer),
    poly-A stretches (every start a modimizer of one k-mer when it is one), and ordinary reads in between so that waves mix k-mers;
    both flag polarities, both element formats, two batches (the second finds the hot k-mers already in the table).  Against the oracle:
    value[], depth[] (saturating at 65 535: modutils.c:26) and index[]."""
    import subprocess, sys, os
    code = r"""
import numpy as np, modimizer_amd as mg
from oracle import pyoracle as po
k, d = 21, 64
sh = mg.seqhashCreate(k, d, 17); oh = po.Hasher(k, d, 17)
rng = np.random.default_rng(9)
hot = rng.integers(0, 4, 400).astype(np.uint8)
reads = []
for r in range(9000):
    u = rng.random()
    if u < 0.55: reads.append(hot)                                                     # one read, thousands of times
    elif u < 0.75: reads.append(np.concatenate([rng.integers(0, 4, 30).astype(np.uint8), np.zeros(int(rng.integers(25, 400)), np.uint8), rng.integers(0, 4, 30).astype(np.uint8)]))   # poly-A inside a read
    elif u < 0.80: reads.append(3 - hot[::-1])                                         # its reverse complement: the same k-mers
    else: reads.append(rng.integers(0, 4, int(rng.integers(21, 800))).astype(np.uint8))
offs = np.concatenate([[0], np.cumsum([len(x) for x in reads])]).astype(np.int64)
b = np.concatenate(reads)
bits = 20
ms = mg.modsetCreate(sh, bits); oms = po.Modset(oh, bits)
half = len(reads) // 2
n = mg.add_sequence_batch(ms, b[:int(offs[half])], offs[:half + 1])
n += mg.add_sequence_batch(ms, b, offs)
t = sum(oms.add_sequence(x) for x in reads[:half]) + sum(oms.add_sequence(x) for x in reads)
mg.check(mg.lib().modsetSyncToHost(ms, 1))
v, dep, _ = mg.modset_arrays(ms)
assert n == t and ms.contents.max == oms.max, (n, t, ms.contents.max, oms.max)
assert np.array_equal(v[1:], oms.values()[1:]) and np.array_equal(dep[1:], oms.depths()[1:])
assert int(dep[1:].max()) == 65535                                                     # the hot k-mers saturate
assert np.array_equal(np.ctypeslib.as_array(ms.contents.index, (1 << bits,)), oms.index_table())
print("hot ok", n, oms.max)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for knobs in ({}, {"MODGPU_FLAG_POLARITY": "0"}, {"MODGPU_FLAG_POLARITY": "1"}, {"MODGPU_PART_PACKED": "0", "MODGPU_FLAG_POLARITY": "1"},
                  {"MODGPU_BUCKET_R": "1024", "MODGPU_BUCKET_T": "256"},
                  # the split of oversize buckets (mgHotPlanKernel / mgHotReduceKernel / mgDedupHotBucket): the poly-A k-mer's bucket takes it with
                  # the defaults above (> 32768 occurrences); "200,64" sends every bucket of more than 200 occurrences through it in chunks of 64,
                  # both polarities and both element formats; the last switches it off (the single-workgroup path as it was)
                  {"MODGPU_HOT_SPLIT": "200,64"}, {"MODGPU_HOT_SPLIT": "200,64", "MODGPU_FLAG_POLARITY": "0"},
                  {"MODGPU_HOT_SPLIT": "200,64", "MODGPU_FLAG_POLARITY": "1", "MODGPU_PART_PACKED": "0"},
                  {"MODGPU_HOT_SPLIT": "3000,1000", "MODGPU_BUCKET_R": "1024", "MODGPU_BUCKET_T": "256"},
                  {"MODGPU_HOT_SPLIT": "2000000000"},
                  {"MODGPU_TABLE_PATH": "direct"}):            # the atomic path: a wave's lanes in one slot add their number once (mgTableInsertKernel)
        env = dict(os.environ, PYTHONPATH=root, **dict({"MODGPU_TABLE_PATH": "bucket"}, **knobs))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "hot ok" in r.stdout, (knobs, r.stdout[-300:], r.stderr[-1500:])


def _revcomp(b):
    return (3 - b[::-1]).astype(np.uint8)


def _mutate(rng, b, rate):
    b = b.copy()
    hit = rng.random(len(b)) < rate
    b[hit] = (b[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
    return b


@pytest.mark.parametrize("path", ["direct", "part", "2 levels"])
@pytest.mark.parametrize("k,w,seed", [(15, 8, 17), (21, 64, 17), (19, 31, 17), (13, 4, 5), (17, 16, 17), (25, 11, 3)])
def test_modmap_randomized_vs_oracle(k, w, seed, path, tmp_path):
    """queryProcess (modmap.c:188-281) on randomized references and reads against the ORACLE's restatement of it
    (oracle/orc_modset.c orcQueryRead, itself pinned to the reference program's golden Q / M lines), byte for byte: references
    with duplicated and triplicated segments (copy-2 / copy-M classes, the second-copy retry of modmap.c:242-254), forward
    and reverse-strand reads, substitutions, chimeric reads, reads stitched from up to 60 pieces (more M blocks than the
    device chain kernel keeps per read: the overflow path), reads that hit nothing, reads shorter than k, empty reads.
    6 parameter sets x 6 references x 7-9 reads = about 290 reads; the reference build's own report lines and arrays
    (index / offset / id / depth / loc / rev, modmap.c:74-134) are compared on the way."""
    L = mg.lib()
    with mg.knobs(FIND_PATH=path):                      # the seeds' lookups by direct probes / through the partitioned path
        _modmap_randomized(L, k, w, seed, tmp_path)


def _modmap_randomized(L, k, w, seed, tmp_path):
    rng = np.random.default_rng(1000 * k + w)
    n_reads_total = n_m_lines = n_overflow = 0
    for trial in range(6):
        # ---- a reference of 1-3 sequences with copies of segments inside and across sequences -----------------------
        n_seq = int(rng.integers(1, 4))
        seqs = [rng.integers(0, 4, int(rng.integers(30_000, 90_000))).astype(np.uint8) for _ in range(n_seq)]
        for _ in range(int(rng.integers(2, 6))):
            src = seqs[int(rng.integers(0, n_seq))]
            ln = int(rng.integers(300, 6000))
            a = int(rng.integers(0, len(src) - ln))
            piece = src[a:a + ln].copy()
            if rng.random() < 0.4:
                piece = _revcomp(piece)
            for _copy in range(int(rng.integers(1, 3))):               # one or two extra copies: copy 2 / copy M
                dst = seqs[int(rng.integers(0, n_seq))]
                b = int(rng.integers(0, len(dst) - ln))
                dst[b:b + ln] = _mutate(rng, piece, 0.002 * rng.random())
        names = ["s%d_%d" % (trial, i) for i in range(n_seq)]
        g = np.concatenate(seqs)
        # ---- reads --------------------------------------------------------------------------------------------------
        def cut(ln):
            s = seqs[int(rng.integers(0, n_seq))]
            ln = min(ln, len(s) - 1)
            a = int(rng.integers(0, len(s) - ln))
            r = s[a:a + ln]
            return _revcomp(r) if rng.random() < 0.5 else r.copy()
        reads = [cut(int(rng.integers(2000, 20000))),                                        # plain
                 _mutate(rng, cut(int(rng.integers(3000, 15000))), 0.03),                       # substitutions
                 np.concatenate([cut(int(rng.integers(1500, 6000))) for _ in range(int(rng.integers(2, 5)))]),   # chimeric
                 np.concatenate([cut(int(rng.integers(700, 2500))) for _ in range(int(rng.integers(20, 61)))]),  # > 16 blocks
                 rng.integers(0, 4, int(rng.integers(500, 5000))).astype(np.uint8),           # hits nothing
                 cut(k - 1) if trial % 2 else np.zeros(0, np.uint8),                           # shorter than k / empty
                 np.concatenate([cut(4000), rng.integers(0, 4, 800).astype(np.uint8), cut(4000)])]               # a gap of junk
        if trial % 3 == 0:
            reads.append(_mutate(rng, np.concatenate([cut(900) for _ in range(25)]), 0.01))
            reads.append(np.zeros(3000, np.uint8))                                             # poly-A
        rnames = ["q%d_%d" % (trial, i) for i in range(len(reads))]
        # ---- oracle -------------------------------------------------------------------------------------------------
        oh = po.Hasher(k, w, seed); oms = po.Modset(oh, 20); oref = po.Reference(oms)
        for nm, s in zip(names, seqs):
            oref.add_sequence(nm, s)
        oref.finish()
        want = "".join(oref.query(nm, r, str(tmp_path / "o.txt"))[0] for nm, r in zip(rnames, reads))
        oa = oref.arrays()
        # ---- the library --------------------------------------------------------------------------------------------
        sh = mg.seqhashCreate(k, w, seed); ms = mg.modsetCreate(sh, 20)
        ref = L.mgReferenceCreate(ms, 1 << 26)
        offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
        cn = (C.c_char_p * n_seq)(*[n.encode() for n in names])
        with mg.CFile(str(tmp_path / "r.txt"), "w") as f:
            assert L.mgReferenceRead(ref, g.ctypes.data, offs.ctypes.data, n_seq, cn, True, f) == 0
        R = C.cast(ref, C.POINTER(mg.MgReference)).contents
        U, occ = ms.contents.max, R.max
        assert U == oms.max and occ == len(oa["index"])
        as_np = lambda p, n: np.ctypeslib.as_array(p, (max(n, 1),))[:n]
        for key, n in (("index", occ), ("offset", occ), ("id", occ), ("rev", occ), ("depth", U + 1), ("loc", U + 1)):
            assert np.array_equal(as_np(getattr(R, key), n), oa[key]), (trial, key)
        assert np.array_equal(as_np(ms.contents.value, U + 1)[1:], oms.values()[1:])
        assert np.array_equal(as_np(ms.contents.info, U + 1)[1:], oms.infos()[1:])           # copy classes, modmap.c:125-129
        qb, qo = util.concat_reads(reads)
        cq = (C.c_char_p * len(reads))(*[n.encode() for n in rnames])
        out = str(tmp_path / "q.txt")
        with mg.CFile(out, "w") as f:
            assert L.mgQueryProcess(ref, qb.ctypes.data, qo.ctypes.data, len(reads), cq, f) == 0
        got = open(out).read()
        assert got == want, (k, w, trial, [x for x in zip(got.splitlines(), want.splitlines()) if x[0] != x[1]][:3])
        n_reads_total += len(reads)
        n_m_lines += sum(l.startswith("M\t") for l in want.splitlines())
        n_overflow += sum(sum(l.startswith("M\t%s\t" % nm) for l in want.splitlines()) > 16 for nm in rnames)
        L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
        oref.close()
    assert n_reads_total >= 42 and n_m_lines > 20
    assert n_overflow >= 1 or w >= 31, "no read with more than 16 M blocks: the overflow path was not exercised"
